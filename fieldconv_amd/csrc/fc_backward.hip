// FieldConv backward: filter-gradient kernel, the reductions, the fp32-MFMA instantiation of the data
// kernel and the mode dispatch (data kernel: fc_backward_kernels.hpp).
#include <stdio.h>
#include "fc_backward_kernels.hpp"
#include "fc_backward_stream.hpp"

namespace fc {

// ---------------------------------------------------------------------------------- filter gradient
// T = 16x16 complex gW tiles owned by each wavefront (ceil(ngw / 16)).
// The complex product H conj(X), H = a + ib, X = c + id, uses three real products instead of four
// (Gauss): k1 = (a+b) c, k2 = a (c+d), k3 = b (c-d); re = k1 - k3 = ac + bd, im = k1 - k2 = bc - ad.
// The X combinations are formed once per tile when the rotated features go to LDS, a+b costs one add
// per fragment element: 12 MFMAs per 16 vertices and tile instead of 16, 12*T accumulator VGPRs.
template <int T>
__global__ __launch_bounds__(kThreads) void fc_backward_filter_kernel(
    const float2* __restrict__ gx_, const float* __restrict__ hdump, float2* __restrict__ ggwp /* [P][F][KP][IP] */,
    const BwdArgs a, const int F, const int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MmaGeom& mg = a.g;
    const int KD = a.KD, KP = mg.KP, IP = mg.MP, I = a.I;
    float* const slab0 = reinterpret_cast<float*>(smem);               // [2][slab_stride]: H of a tile, [16 vertices][KD] interleaved (re, im)
    // rotated features of the tile, rows of kXtStride = 20 floats (16 vertices + pad): the 16-byte B-fragment
    // reads of 16 consecutive rows then fall into 16 distinct bank groups
    float* const xtr = slab0 + 2 * a.slab_stride;                      // [IP][20]  c
    float* const xts = xtr + IP * kXtStride;                           // [IP][20]  c + d
    float* const xtd = xts + IP * kXtStride;                           // [IP][20]  c - d

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y;
    const int m = f - B;
    const int fr = lane & 15, fq = lane >> 4;

    for (int idx = tid; idx < 3 * IP * kXtStride; idx += kThreads) xtr[idx] = 0.f;

    // my gW tiles: u = wave + 16 n  ->  (row tile rt over k = (r,o), column tile ct over i)
    int gw_h[T], gw_x[T];       // wave-uniform LDS offsets, -1 when the slot is unused
#pragma unroll
    for (int n = 0; n < T; ++n) {
        const int u = wave + kWaves * n;
        const int rt = u / mg.NMT, ct = u - rt * mg.NMT;
        gw_h[n] = (u < a.ngw) ? rt * 32 : -1;            // floats: 16 complex entries per row tile
        gw_x[n] = ct * 16 * kXtStride;
    }
    const int h_lane = (4 * fq) * KD + 2 * fr;    // A fragment: H[vertex 4fq+s][k = rt*16 + fr], one 8-byte (re, im) read
    const int x_lane = fr * kXtStride + 4 * fq;   // B fragment: xt[i = ct*16 + fr][vertex 4fq..4fq+3]

    f32x4 k1[T], k2[T], k3[T];
#pragma unroll
    for (int n = 0; n < T; ++n) { k1[n] = f32x4{0.f, 0.f, 0.f, 0.f}; k2[n] = k1[n]; k3[n] = k1[n]; }

    const int npieces = a.slab_stride / 256;      // 1 KiB DMA pieces per slab
    auto dma_slab = [&](const int tile, const int buf) {
        const float* src = hdump + ((size_t)tile * F + f) * a.slab_stride;
        if (!(a.dbg & 16))
        for (int p = wave; p < npieces; p += kWaves)
            lds_dma16_untracked(src + p * 256 + lane * 4, slab0 + buf * a.slab_stride + p * 256);
    };

    if (blockIdx.x < a.ntiles) dma_slab(blockIdx.x, 0);
    float2 xv = make_float2(0.f, 0.f);
    {
        const int j = item_vertex(blockIdx.x, wave, a.nv_full, a.parts_log2, a.N);
        if (blockIdx.x < a.ntiles && j < a.N && lane < I) xv = gx_[(size_t)j * I + lane];
    }
    int buf = 0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, buf ^= 1) {
        // rotated feature xt_f of my source row (B operand)
        const float2 xt = cmul(xv, unit_power(unit_conj(xv), m));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of slab `buf` have landed
        __syncthreads();            // everyone's have; the previous tile's reads are done
        if (lane < IP) {
            xtr[lane * kXtStride + wave] = xt.x;
            xts[lane * kXtStride + wave] = xt.x + xt.y;
            xtd[lane * kXtStride + wave] = xt.x - xt.y;
        }
        const int tn = tile + gridDim.x;
        if (tn < a.ntiles) {
            dma_slab(tn, buf ^ 1);
            const int jn = item_vertex(tn, wave, a.nv_full, a.parts_log2, a.N);
            xv = (jn < a.N && lane < I) ? gx_[(size_t)jn * I + lane] : make_float2(0.f, 0.f);
        }
        // the xt stores must be visible to every wavefront; LDS only, the DMA just issued stays in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        const float* hsl = slab0 + buf * a.slab_stride;
        // gW += H^T . conj(xt)
        if (!(a.dbg & 4)) {
#pragma unroll
            for (int n = 0; n < T; ++n) {
                if (gw_h[n] >= 0) {
                    const float4 c4 = *reinterpret_cast<const float4*>(xtr + gw_x[n] + x_lane);
                    const float4 s4 = *reinterpret_cast<const float4*>(xts + gw_x[n] + x_lane);
                    const float4 d4 = *reinterpret_cast<const float4*>(xtd + gw_x[n] + x_lane);
                    const float* ha = hsl + gw_h[n] + h_lane;
                    const float2 h0 = *reinterpret_cast<const float2*>(ha), h1 = *reinterpret_cast<const float2*>(ha + KD);
                    const float2 h2 = *reinterpret_cast<const float2*>(ha + 2 * KD), h3 = *reinterpret_cast<const float2*>(ha + 3 * KD);
                    const float a0 = h0.x, a1 = h1.x, a2 = h2.x, a3 = h3.x;
                    const float b0 = h0.y, b1 = h1.y, b2 = h2.y, b3 = h3.y;
                    k1[n] = mfma16(a0 + b0, c4.x, k1[n]); k2[n] = mfma16(a0, s4.x, k2[n]); k3[n] = mfma16(b0, d4.x, k3[n]);
                    k1[n] = mfma16(a1 + b1, c4.y, k1[n]); k2[n] = mfma16(a1, s4.y, k2[n]); k3[n] = mfma16(b1, d4.y, k3[n]);
                    k1[n] = mfma16(a2 + b2, c4.z, k1[n]); k2[n] = mfma16(a2, s4.z, k2[n]); k3[n] = mfma16(b2, d4.z, k3[n]);
                    k1[n] = mfma16(a3 + b3, c4.w, k1[n]); k2[n] = mfma16(a3, s4.w, k2[n]); k3[n] = mfma16(b3, d4.w, k3[n]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // flush my gW partial
#pragma unroll
    for (int n = 0; n < T; ++n) {
        if (gw_h[n] >= 0) {
            const int ct16 = gw_x[n] / kXtStride;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int k = gw_h[n] / 2 + 4 * fq + jj;
                const int i = ct16 + fr;
                ggwp[(((size_t)blockIdx.x * F + f) * KP + k) * IP + i] = make_float2(k1[n][jj] - k3[n][jj], k1[n][jj] - k2[n][jj]);
            }
        }
    }
}

// ------------------------------------------------ filter gradient on half-precision operands
// The same contraction gW[k,i] = sum_v H[v,k] conj(xt[v,i]) on v_mfma_f32_16x16x16_f16 (k dimension = the 16
// vertices of a tile), from the same fp32 slabs.  The fp32 matrix pipe bounds the kernel above; here
// every operand is carried as two halves (fc_tile.hpp, split mode) and a product is hi*hi + hi*lo + lo*hi:
//   first operand : each wavefront converts its vertex's row of the slab once per tile, scaled by the
//                   power-of-two s_v the data kernel used for that vertex, into an LDS image
//                   [vertex][plane][k]; the k-major fragments come out of it with ds_read_b64_tr_b16;
//   second operand: xt[v,i] / s_v -- a scale per vertex cannot be factored out of a sum over vertices, so it
//                   moves to the other factor -- times one power-of-two scale t[i] per column and tile,
//                   split into halves, planes [i][vertex].
// All scales arrive with the slab (BwdArgs::tails), so the conversion has no reduction and no extra barrier.
// The tile's products start from zero and are added to the fp32 running sums with the factor 1/t[i].
// Two variants: fc_backward_filter_half_kernel stages the fp32 slabs in LDS (LDS-DMA, double-buffered; one image, two
// barriers per slab; FC_FILTER2=0), fc_backward_filter_half2_kernel (default, below it) converts straight from registers
// loaded one slab ahead into one of two images, with one barrier per slab.
__device__ __forceinline__ f32x4 mfma16h(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
}

template <int T>
__global__ __launch_bounds__(kThreads) void fc_backward_filter_half_kernel(
    const float2* __restrict__ gx_, const float* __restrict__ hdump, float2* __restrict__ ggwp /* [P][F][KP][IP] */,
    const BwdArgs a, const int F, const int B, const int KV /* valid k entries, R*O */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MmaGeom& mg = a.g;
    const int KD = a.KD, KP = mg.KP, IP = mg.MP, I = a.I;
    const int KSI = filter_image_stride(KP);
    const int xplane = IP * kXbStride;
    float* const slab0 = reinterpret_cast<float*>(smem);               // [2][slab_stride]: H of a tile, fp32, + scales (DMA target)
    lds_f16* const img = (lds_f16*)(slab0 + 2 * a.slab_stride);        // [16 vertices][re_hi, re_lo, im_hi, im_lo][KP] halves, row stride KSI
    lds_f16* const xb = img + kTile * KSI;                             // [c_hi, c_lo, d_hi, d_lo, -d_hi, -d_lo][IP][kXbStride] halves
    lds_f16* const zeros = xb + 6 * IP * kXbStride;                    // 16 bytes of zeros: the lower half of the [lo; 0] fragments

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y;
    const int m = f - B;
    const int fr = lane & 15, fq = lane >> 4;

    for (int idx = tid; idx < (kTile * KSI + 6 * xplane) / 2 + 4; idx += kThreads) slab0[2 * a.slab_stride + idx] = 0.f;

    // my gW tiles all lie in ONE column tile (i0 = 16 * (wave % NMT)), so the second-operand fragments are read once per
    // tile; the row tiles of that column are dealt round-robin to the wavefronts that share it
    const int my_ct = wave % mg.NMT, my_idx = wave / mg.NMT;
    const int ct_waves = (kWaves - my_ct + mg.NMT - 1) / mg.NMT;
    const int i0 = my_ct * 16;
    int gw_h[T];                // first k of my row tiles; -1: unused slot
#pragma unroll
    for (int n = 0; n < T; ++n) {
        const int rt = my_idx + n * ct_waves;
        gw_h[n] = (rt * 16 < KP) ? rt * 16 : -1;
    }
    // The MFMA's 32 k entries are the 16 vertices' hi halves followed by their lo halves (first operand) against
    // [hi; hi] and [lo; 0] of the second: (A_hi | A_lo)(B_hi; B_hi) + (A_hi | A_lo)(B_lo; 0) = hi*hi + lo*hi + hi*lo, two
    // full-rate v_mfma_f32_16x16x32_f16 per real product.  Lane groups 0/1 carry vertices 0-7 / 8-15 of the hi planes,
    // groups 2/3 the same vertices of the lo planes.
    // A fragment: lane 4q+p of a lane group addresses vertex (vb + q), entries k0+4p .. +3 of its plane, and receives entry
    // k0 + lane%16 of the vertices vb .. vb+3; two reads (vb = 8*(g&1), +4) make the group's eight k entries.
    const int a_lane = (8 * (fq & 1) + ((lane & 15) >> 2)) * KSI + 4 * (lane & 3) + (fq >= 2 ? KP : 0);
    const int b_lane = fr * kXbStride + 8 * (fq & 1);    // B fragment: plane[i = i0 + fr][vertices 8*(g&1) .. +7]
    const bool upper = fq >= 2;

    f32x4 gre[T], gim[T];
#pragma unroll
    for (int n = 0; n < T; ++n) { gre[n] = f32x4{0.f, 0.f, 0.f, 0.f}; gim[n] = gre[n]; }

    const int npieces = a.slab_stride / 256;      // 1 KiB DMA pieces per slab
    auto dma_slab = [&](const int tile, const int buf) {
        const float* src = hdump + ((size_t)tile * F + f) * a.slab_stride;
        for (int p = wave; p < npieces; p += kWaves)
            lds_dma16_untracked(src + p * 256 + lane * 4, slab0 + buf * a.slab_stride + p * 256);
    };

    if (blockIdx.x < a.ntiles) dma_slab(blockIdx.x, 0);
    float2 xv = make_float2(0.f, 0.f);
    {
        const int j = item_vertex(blockIdx.x, wave, a.nv_full, a.parts_log2, a.N);
        if (blockIdx.x < a.ntiles && j < a.N && lane < I) xv = gx_[(size_t)j * I + lane];
    }
    const int nchunk = KP / 4;                     // chunks of four k entries; a lane converts chunks lane and lane + 64
    int buf = 0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, buf ^= 1) {
        const float2 xt = cmul(xv, unit_power(unit_conj(xv), m));     // rotated feature of my source row
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // my pieces of slab `buf` have landed
        __syncthreads();                                               // everyone's have; the previous tile's MFMAs are done
        const int tn = tile + gridDim.x;
        if (tn < a.ntiles) {
            dma_slab(tn, buf ^ 1);
            const int jn = item_vertex(tn, wave, a.nv_full, a.parts_log2, a.N);
            xv = (jn < a.N && lane < I) ? gx_[(size_t)jn * I + lane] : make_float2(0.f, 0.f);
        } else {
            xv = make_float2(0.f, 0.f);
        }
        const float* const sl = slab0 + buf * a.slab_stride;
        const float* const tail = sl + a.slab_floats;       // [16] s_v, [16] 1/s_v, [IP] t, [IP] 1/t
        {   // ---- my vertex's row of H: scale, split, store plane-major
            const float* row = sl + wave * KD;
            const float s = tail[wave];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int c = lane + 64 * cc;
                if (c < nchunk) {
                    float4 v0 = *reinterpret_cast<const float4*>(row + 8 * c);          // (re, im) of k = 4c, 4c+1
                    float4 v1 = *reinterpret_cast<const float4*>(row + 8 * c + 4);      // k = 4c+2, 4c+3
                    // entries past the R*O valid ones were never written by the data kernel
                    if (4 * c + 0 >= KV) { v0.x = 0.f; v0.y = 0.f; }
                    if (4 * c + 1 >= KV) { v0.z = 0.f; v0.w = 0.f; }
                    if (4 * c + 2 >= KV) { v1.x = 0.f; v1.y = 0.f; }
                    if (4 * c + 3 >= KV) { v1.z = 0.f; v1.w = 0.f; }
                    f16x2 h0, l0, h1, l1, h2, l2, h3, l3;
                    split_halves2(f32x2{v0.x, v0.y}, s, h0, l0);
                    split_halves2(f32x2{v0.z, v0.w}, s, h1, l1);
                    split_halves2(f32x2{v1.x, v1.y}, s, h2, l2);
                    split_halves2(f32x2{v1.z, v1.w}, s, h3, l3);
                    lds_f16* p = img + wave * KSI + 4 * c;
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(3))) h4 lds_h4;
                    *(lds_h4*)(p) = h4{h0.x, h1.x, h2.x, h3.x};                 // re_hi
                    *(lds_h4*)(p + KP) = h4{l0.x, l1.x, l2.x, l3.x};            // re_lo
                    *(lds_h4*)(p + 2 * KP) = h4{h0.y, h1.y, h2.y, h3.y};        // im_hi
                    *(lds_h4*)(p + 3 * KP) = h4{l0.y, l1.y, l2.y, l3.y};        // im_lo
                }
            }
        }
        if (lane < IP) {   // ---- second operand: xt / s_v * t[i], halves, planes [i][vertex]
            f16x2 hi, lo;
            split_halves2(f32x2{xt.x, xt.y}, tail[kTile + wave] * tail[2 * kTile + lane], hi, lo);
            lds_f16* p = xb + lane * kXbStride + wave;
            p[0] = hi.x;
            p[xplane] = lo.x;
            p[2 * xplane] = hi.y;
            p[3 * xplane] = lo.y;
            p[4 * xplane] = -hi.y;
            p[5 * xplane] = -lo.y;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        if (!(a.dbg & 4)) {
            // second operand, once per tile: [hi; hi] from the hi planes in every lane group, [lo; 0] from the lo planes in
            // groups 0/1 and from the zero block in groups 2/3
            const lds_f16* bp = xb + i0 * kXbStride + b_lane;
            const lds_f16* bl = upper ? zeros : bp + xplane;
            const int lstep = upper ? 0 : 2 * xplane;
            const u32x4 c_hh = *reinterpret_cast<lds_u32x4*>(bp), c_l0 = *reinterpret_cast<lds_u32x4*>(bl);
            const u32x4 d_hh = *reinterpret_cast<lds_u32x4*>(bp + 2 * xplane), d_l0 = *reinterpret_cast<lds_u32x4*>(bl + lstep);
            const u32x4 nd_hh = *reinterpret_cast<lds_u32x4*>(bp + 4 * xplane), nd_l0 = *reinterpret_cast<lds_u32x4*>(bl + 2 * lstep);
            const float it = tail[2 * kTile + IP + i0 + fr];
#pragma unroll
            for (int n = 0; n < T; ++n) {
                if (gw_h[n] >= 0) {
                    // first operand: (hi | lo) of the real and of the imaginary part of H^T
                    const lds_f16* ap = img + a_lane + gw_h[n];
                    u32x4 are, aim;
                    {
                        const u32x2 r0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap)));
                        const u32x2 r1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 4 * KSI)));
                        const u32x2 i0_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 2 * KP)));
                        const u32x2 i1_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 2 * KP + 4 * KSI)));
                        are = u32x4{r0.x, r0.y, r1.x, r1.y};
                        aim = u32x4{i0_.x, i0_.y, i1_.x, i1_.y};
                    }
                    // H conj(X), H = a + ib, X = c + id:  re = a c + b d,  im = b c - a d
                    f32x4 re = {0.f, 0.f, 0.f, 0.f}, im = re;
                    re = mfma32h(are, c_l0, re);  re = mfma32h(are, c_hh, re);
                    re = mfma32h(aim, d_l0, re);  re = mfma32h(aim, d_hh, re);
                    im = mfma32h(aim, c_l0, im);  im = mfma32h(aim, c_hh, im);
                    im = mfma32h(are, nd_l0, im); im = mfma32h(are, nd_hh, im);
                    gre[n] += re * it;
                    gim[n] += im * it;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // flush my gW partial
#pragma unroll
    for (int n = 0; n < T; ++n) {
        if (gw_h[n] >= 0) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int k = gw_h[n] + 4 * fq + jj;
                const int i = i0 + fr;
                ggwp[(((size_t)blockIdx.x * F + f) * KP + k) * IP + i] = make_float2(gre[n][jj], gim[n][jj]);
            }
        }
    }
}

// HALVES: the kept slabs hold the data kernel's own halves (BwdArgs::dump_halves): the rows are regrouped, not converted, and the
// registers that frees hold a second slab row in flight.
template <int T, bool HALVES>
__global__ __launch_bounds__(kThreads) void fc_backward_filter_half2_kernel(
    const float2* __restrict__ gx_, const float* __restrict__ hdump, float2* __restrict__ ggwp /* [P][F][KP][IP] */,
    const BwdArgs a, const int F, const int B, const int KV /* valid k entries, R*O */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MmaGeom& mg = a.g;
    const int KD = a.KD, KP = mg.KP, IP = mg.MP, I = a.I;
    const int KSI = filter_image_stride(KP);
    const int xplane = IP * kXbStride;
    // Two images and two second-operand blocks: the rows of slab t+1 are converted (straight from registers, loaded from
    // memory one slab ahead) while slower wavefronts still run the MFMAs of slab t -- one barrier per slab, no fp32 copy
    // of H in LDS.
    lds_f16* const img0 = (lds_f16*)smem;                              // [2][16 vertices][re_hi, re_lo, im_hi, im_lo][KP] halves, row stride KSI
    lds_f16* const xb0 = img0 + 2 * kTile * KSI;                       // [2][c_hi, c_lo, d_hi, d_lo, -d_hi, -d_lo][IP][kXbStride] halves
    lds_f16* const zeros = xb0 + 2 * 6 * IP * kXbStride;               // 16 bytes of zeros: the lower half of the [lo; 0] fragments

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y;
    const int m = f - B;
    const int fr = lane & 15, fq = lane >> 4;

    for (int idx = tid; idx < (2 * kTile * KSI + 2 * 6 * xplane) / 2 + 4; idx += kThreads) reinterpret_cast<float*>(smem)[idx] = 0.f;
    __syncthreads();

    // my gW tiles all lie in ONE column tile (i0 = 16 * (wave % NMT)), so the second-operand fragments are read once per
    // tile; the row tiles of that column are dealt round-robin to the wavefronts that share it
    const int my_ct = wave % mg.NMT, my_idx = wave / mg.NMT;
    const int ct_waves = (kWaves - my_ct + mg.NMT - 1) / mg.NMT;
    const int i0 = my_ct * 16;
    int gw_h[T];                // first k of my row tiles; -1: unused slot
#pragma unroll
    for (int n = 0; n < T; ++n) {
        const int rt = my_idx + n * ct_waves;
        gw_h[n] = (rt * 16 < KP) ? rt * 16 : -1;
    }
    // The MFMA's 32 k entries are the 16 vertices' hi halves followed by their lo halves (first operand) against [hi; hi] of
    // the second: (A_hi | A_lo)(B_hi; B_hi) = hi*hi + lo*hi; the hi*lo products of the TWO real products that make one
    // component share a third instruction, (a_hi | b_hi)[c_lo; d_lo] -- three full-rate v_mfma_f32_16x16x32_f16 per component,
    // every k entry used.  Lane groups 0/1 carry vertices 0-7 / 8-15 of the first block, groups 2/3 the same vertices of the
    // second.
    // A fragment: lane 4q+p of a lane group addresses vertex (vb + q), entries k0+4p .. +3 of its plane, and receives entry
    // k0 + lane%16 of the vertices vb .. vb+3; two reads (vb = 8*(g&1), +4) make the group's eight k entries.
    const int a_lane = (8 * (fq & 1) + ((lane & 15) >> 2)) * KSI + 4 * (lane & 3) + (fq >= 2 ? KP : 0);
    const int a_lane_hi = (8 * (fq & 1) + ((lane & 15) >> 2)) * KSI + 4 * (lane & 3) + (fq >= 2 ? 2 * KP : 0);     // (re_hi | im_hi)
    const int b_lane = fr * kXbStride + 8 * (fq & 1);    // B fragment: plane[i = i0 + fr][vertices 8*(g&1) .. +7]
    const bool upper = fq >= 2;

    f32x4 gre[T], gim[T];
#pragma unroll
    for (int n = 0; n < T; ++n) { gre[n] = f32x4{0.f, 0.f, 0.f, 0.f}; gim[n] = gre[n]; }

    const int nchunk = KP / 4;                     // chunks of four k entries; a lane holds chunks lane and lane + 64 of its vertex's row
    // registers one slab ahead: my vertex's row of H (fp32) and the scales behind the slab
    // Registers one slab ahead: my vertex's row of H and the scales behind the slab.  (A second set, two slabs ahead, does not
    // fit the 128 registers -- and would not pay: with the rows of ONE tile re-read from L2 all the time, FC_DEBUG_BWD=16, the
    // kernel takes 73.0 instead of 74.5 us; it does not wait for HBM.)
    struct Ahead {
        float4 pv[2][2];
        float ps, pinv, pt, pit;
        float2 xv;
    };
    Ahead ahead0;
    auto prefetch = [&](const int tile, Ahead& q) {
        const float* sl = hdump + ((size_t)((a.dbg & 16) ? (int)blockIdx.x : tile) * F + f) * a.slab_stride;      // (development: bit 4 -- always my first tile's rows, from L2)
        const float* row = sl + wave * KD;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = lane + 64 * cc;
            if (c < nchunk) {
                q.pv[cc][0] = *reinterpret_cast<const float4*>(row + 8 * c);          // (re, im) of k = 4c, 4c+1
                q.pv[cc][1] = *reinterpret_cast<const float4*>(row + 8 * c + 4);      // k = 4c+2, 4c+3
            }
        }
        const float* tail = sl + a.slab_floats;             // [16] s_v, [16] 1/s_v, [IP] t, [IP] 1/t
        q.ps = tail[wave];
        q.pinv = tail[kTile + wave];
        q.pt = lane < IP ? tail[2 * kTile + lane] : 1.f;
        q.pit = tail[2 * kTile + IP + i0 + fr];
        const int j = item_vertex(tile, wave, a.nv_full, a.parts_log2, a.N);
        q.xv = (j < a.N && lane < I) ? gx_[(size_t)j * I + lane] : make_float2(0.f, 0.f);
    };
    {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) { ahead0.pv[cc][0] = z4; ahead0.pv[cc][1] = z4; }
        ahead0.ps = ahead0.pinv = ahead0.pt = ahead0.pit = 1.f;
        ahead0.xv = make_float2(0.f, 0.f);
    }
    const int tile_step = gridDim.x;
    Stamper stamp{(a.stamps && a.stamp_who == 2 && blockIdx.x == 0 && blockIdx.y == 0) ? a.stamps + wave * 256 : nullptr, 0};
    stamp.realtime(29);
    stamp(28);
    if ((int)blockIdx.x < a.ntiles) prefetch(blockIdx.x, ahead0);
    auto one_tile = [&](const int tile, Ahead& q, const int buf) {
        stamp(10);
        lds_f16* const img = img0 + buf * kTile * KSI;
        lds_f16* const xb = xb0 + buf * 6 * xplane;
        const float2 xt = cmul(q.xv, unit_power(unit_conj(q.xv), m));     // rotated feature of my source row
        const float it = q.pit;
        {   // ---- my vertex's row of H: scale, split, store plane-major
            const float s = q.ps;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int c = lane + 64 * cc;
                if (c < nchunk) {
                    float4 v0 = q.pv[cc][0], v1 = q.pv[cc][1];
                    // entries past the R*O valid ones were never written by the data kernel
                    if (4 * c + 0 >= KV) { v0.x = 0.f; v0.y = 0.f; }
                    if (4 * c + 1 >= KV) { v0.z = 0.f; v0.w = 0.f; }
                    if (4 * c + 2 >= KV) { v1.x = 0.f; v1.y = 0.f; }
                    if (4 * c + 3 >= KV) { v1.z = 0.f; v1.w = 0.f; }
                    lds_f16* p = img + wave * KSI + 4 * c;
                    if constexpr (HALVES) {
                        // the slab holds the data kernel's own halves, (hi.re, hi.im | lo.re, lo.im) per entry: regroup plane-major
                        const uint32_t k0h = __float_as_uint(v0.x), k0l = __float_as_uint(v0.y), k1h = __float_as_uint(v0.z), k1l = __float_as_uint(v0.w);
                        const uint32_t k2h = __float_as_uint(v1.x), k2l = __float_as_uint(v1.y), k3h = __float_as_uint(v1.z), k3l = __float_as_uint(v1.w);
                        constexpr uint32_t kLow = 0x05040100u, kHigh = 0x07060302u;      // (b.lo16, a.lo16) / (b.hi16, a.hi16) of perm(a, b)
                        typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
                        *(lds_u32x2*)(p) = u32x2{__builtin_amdgcn_perm(k1h, k0h, kLow), __builtin_amdgcn_perm(k3h, k2h, kLow)};                 // re_hi
                        *(lds_u32x2*)(p + KP) = u32x2{__builtin_amdgcn_perm(k1l, k0l, kLow), __builtin_amdgcn_perm(k3l, k2l, kLow)};            // re_lo
                        *(lds_u32x2*)(p + 2 * KP) = u32x2{__builtin_amdgcn_perm(k1h, k0h, kHigh), __builtin_amdgcn_perm(k3h, k2h, kHigh)};      // im_hi
                        *(lds_u32x2*)(p + 3 * KP) = u32x2{__builtin_amdgcn_perm(k1l, k0l, kHigh), __builtin_amdgcn_perm(k3l, k2l, kHigh)};      // im_lo
                    } else {
                        f16x2 h0, l0, h1, l1, h2, l2, h3, l3;
                        split_halves2(f32x2{v0.x, v0.y}, s, h0, l0);
                        split_halves2(f32x2{v0.z, v0.w}, s, h1, l1);
                        split_halves2(f32x2{v1.x, v1.y}, s, h2, l2);
                        split_halves2(f32x2{v1.z, v1.w}, s, h3, l3);
                        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                        typedef __attribute__((address_space(3))) h4 lds_h4;
                        *(lds_h4*)(p) = h4{h0.x, h1.x, h2.x, h3.x};                 // re_hi
                        *(lds_h4*)(p + KP) = h4{l0.x, l1.x, l2.x, l3.x};            // re_lo
                        *(lds_h4*)(p + 2 * KP) = h4{h0.y, h1.y, h2.y, h3.y};        // im_hi
                        *(lds_h4*)(p + 3 * KP) = h4{l0.y, l1.y, l2.y, l3.y};        // im_lo
                    }
                }
            }
        }
        stamp(1);
        if (lane < IP) {   // ---- second operand: xt / s_v * t[i], halves, planes [i][vertex]
            f16x2 hi, lo;
            split_halves2(f32x2{xt.x, xt.y}, q.pinv * q.pt, hi, lo);
            lds_f16* p = xb + lane * kXbStride + wave;
            p[0] = hi.x;
            p[xplane] = lo.x;
            p[2 * xplane] = hi.y;
            p[3 * xplane] = lo.y;
            p[4 * xplane] = -hi.y;
            p[5 * xplane] = -lo.y;
        }
        if (tile + tile_step < a.ntiles) prefetch(tile + tile_step, q);      // the next slab's rows fly during the MFMAs below
        stamp(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stamp(2);

        if (!(a.dbg & 4)) {
            // second operand, once per tile.  The MFMA's 32 k entries are two blocks of the 16 vertices; with H = a + ib and
            // X = c + id in halves, re = a c + b d takes THREE instructions: (a_hi | a_lo)[c_hi; c_hi] + (b_hi | b_lo)[d_hi; d_hi] +
            // (a_hi | b_hi)[c_lo; d_lo] -- the two hi*lo products share one -- and im = b c - a d likewise with [-d_lo; c_lo].
            const lds_f16* bp = xb + i0 * kXbStride + b_lane;
            const u32x4 c_hh = *reinterpret_cast<lds_u32x4*>(bp);
            const u32x4 d_hh = *reinterpret_cast<lds_u32x4*>(bp + 2 * xplane);
            const u32x4 nd_hh = *reinterpret_cast<lds_u32x4*>(bp + 4 * xplane);
            const u32x4 lo_re = *reinterpret_cast<lds_u32x4*>(bp + (upper ? 3 : 1) * xplane);      // [c_lo; d_lo]
            const u32x4 lo_im = *reinterpret_cast<lds_u32x4*>(bp + (upper ? 1 : 5) * xplane);      // [-d_lo; c_lo]
#pragma unroll
            for (int n = 0; n < T; ++n) {
                if (gw_h[n] >= 0) {
                    // first operand: (hi | lo) of the real and of the imaginary part of H^T, and (re_hi | im_hi)
                    const lds_f16* ap = img + a_lane + gw_h[n];
                    const lds_f16* ah = img + a_lane_hi + gw_h[n];
                    u32x4 are, aim, ahi;
                    {
                        const u32x2 r0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap)));
                        const u32x2 r1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 4 * KSI)));
                        const u32x2 i0_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 2 * KP)));
                        const u32x2 i1_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 2 * KP + 4 * KSI)));
                        const u32x2 h0_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ah)));
                        const u32x2 h1_ = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ah + 4 * KSI)));
                        are = u32x4{r0.x, r0.y, r1.x, r1.y};
                        aim = u32x4{i0_.x, i0_.y, i1_.x, i1_.y};
                        ahi = u32x4{h0_.x, h0_.y, h1_.x, h1_.y};
                    }
                    // H conj(X), H = a + ib, X = c + id:  re = a c + b d,  im = b c - a d
                    f32x4 re = {0.f, 0.f, 0.f, 0.f}, im = re;
                    re = mfma32h(ahi, lo_re, re);  im = mfma32h(ahi, lo_im, im);
                    re = mfma32h(are, c_hh, re);   im = mfma32h(aim, c_hh, im);
                    re = mfma32h(aim, d_hh, re);   im = mfma32h(are, nd_hh, im);
                    gre[n] += re * it;
                    gim[n] += im * it;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (kDevSwitches) {
            // Development only (FC_DEBUG_BWD bit 5; never in the product library): what would it cost this kernel to ALSO contract the slab
            // with the filter, gxt[v,i] = sum_k H[v,k] conj(W_f[k,i]) -- the "H-streaming" restructure of the backward pass?  A
            // cost prototype: wavefronts 0 .. NMT-1 each run the full-K product of one 16-column tile of gxt -- the slab image as first
            // operand (row-major fragments, four planes), the OTHER image buffer standing in for a filter image resident in LDS (same
            // reads, same banks; no room for a real one beside two slab images) -- three matrix instructions per real product pair, and
            // store their tile.  Same instruction, LDS and store volume as the real thing; the values are meaningless.
            if ((a.dbg & 32) && wave < mg.NMT) {
                const lds_f16* arow = img + fr * KSI + 8 * fq;
                const lds_f16* brow = img0 + (buf ^ 1) * kTile * KSI + fr * KSI + 8 * fq;
                f32x4 xr = {0.f, 0.f, 0.f, 0.f}, xi = xr;
                for (int kb = 0; kb < KP / 32; ++kb) {
                    const u32x4 arh = *reinterpret_cast<lds_u32x4*>(arow + kb * 32), arl = *reinterpret_cast<lds_u32x4*>(arow + kb * 32 + KP);
                    const u32x4 aih = *reinterpret_cast<lds_u32x4*>(arow + kb * 32 + 2 * KP), ail = *reinterpret_cast<lds_u32x4*>(arow + kb * 32 + 3 * KP);
                    const u32x4 brh = *reinterpret_cast<lds_u32x4*>(brow + kb * 32), brl = *reinterpret_cast<lds_u32x4*>(brow + kb * 32 + KP);
                    const u32x4 bih = *reinterpret_cast<lds_u32x4*>(brow + kb * 32 + 2 * KP), bil = *reinterpret_cast<lds_u32x4*>(brow + kb * 32 + 3 * KP);
                    xr = mfma32h(arh, brh, xr); xr = mfma32h(arh, brl, xr); xr = mfma32h(arl, brh, xr);
                    xr = mfma32h(aih, bih, xr); xr = mfma32h(aih, bil, xr); xr = mfma32h(ail, bih, xr);
                    xi = mfma32h(aih, brh, xi); xi = mfma32h(aih, brl, xi); xi = mfma32h(ail, brh, xi);
                    xi = mfma32h(arh, bih, xi); xi = mfma32h(arh, bil, xi); xi = mfma32h(arl, bih, xi);
                }
                // a tile of gxt per (slab, column tile): 16 x 16 complex numbers (written over the partials' buffer, wrapped: this mode's
                // results are not used)
                const size_t cap = (size_t)gridDim.x * F * KP * IP;
                const size_t base = (((size_t)tile * F + f) * mg.NMT + wave) * 256 % (cap - 256);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) ggwp[base + (4 * fq + jj) * 16 + fr] = make_float2(xr[jj] * it, xi[jj] * it);
            }
        }
    
    };
    int buf = 0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += tile_step, buf ^= 1) {
        one_tile(tile, ahead0, buf);
        stamp(3);
    }
    stamp(30);
    stamp.realtime(31);

    // flush my gW partial
#pragma unroll
    for (int n = 0; n < T; ++n) {
        if (gw_h[n] >= 0) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int k = gw_h[n] + 4 * fq + jj;
                const int i = i0 + fr;
                ggwp[(((size_t)blockIdx.x * F + f) * KP + k) * IP + i] = make_float2(gre[n][jj], gim[n][jj]);
            }
        }
    }
}

// gw_eff[o][i][r][f] = 1/F sum_p gwp[p][f][r*O+o][i]
__global__ void fc_reduce_gw_kernel(const float2* __restrict__ gwp, float2* __restrict__ gw, int P, int F, int R,
                                    int O, int I, int KP, int IP, int pairs /* 1: k = dump_k(r, o); 2: k = o*R + r */) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // over (f, k<R*O, i<I), i fastest
    const int total = F * R * O * I;
    if (idx >= total) return;
    const int i = idx % I;
    const int k = (idx / I) % (R * O);
    const int f = idx / (I * R * O);
    // fixed summation order, eight independent loads in flight
    float2 s = make_float2(0.f, 0.f);
    const size_t stride = (size_t)F * KP * IP;
    const float2* src = gwp + ((size_t)f * KP + k) * IP + i;
    int p = 0;
    for (; p + 8 <= P; p += 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(p + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; }
    }
    for (; p < P; ++p) {
        const float2 v = src[(size_t)p * stride];
        s.x += v.x;
        s.y += v.y;
    }
    const float sc = 1.f / (float)F;
    int r = k / O, o = k - r * O;
    if (pairs == 2) {           // o-major (the streaming arrangement)
        o = k / R;
        r = k - o * R;
    } else if (pairs && k < (R >> 1) * 2 * O) {
        const int pr = k / (2 * O), rem = k - pr * 2 * O;
        r = 2 * pr + (rem & 1);
        o = rem >> 1;
    }
    gw[(((size_t)o * I + i) * R + r) * F + f] = make_float2(s.x * sc, s.y * sc);
}

// ------------------------------------------------------------------------------------ H-streaming arrangement (fc_backward_stream.hpp)
template <int R, int B>
static int launch_gather(const float2* gy, const float* rec, const fc_csr* g, const float* wpk, char* hrec, const StreamArgs& a,
                         const StreamPlan& p, hipStream_t stream) {
    auto kern = fc_backward_gather_kernel<R, B>;
    const size_t lds = (size_t)kWaves * kRingChunks * 1024;
    const int grid = p.ntiles < num_cus() ? p.ntiles : num_cus();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, gy, rec, g->rowptr, g->runs, wpk, hrec, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

template <int T, int KPT = 0, int IT = 0>
static int launch_stream(const float2* x, const char* hrec, const float* wpk, float2* gwp, float2* gxt, const StreamArgs& a,
                         const StreamPlan& p, hipStream_t stream) {
    auto kern = fc_backward_stream_kernel<T, KPT, IT>;
    static bool lds_ok[kMaxDevices] = {};
    if (!allow_full_lds(reinterpret_cast<const void*>(kern), p.lds, lds_ok)) return FC_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(p.P, p.FS), dim3(kThreads), p.lds, stream, x, hrec, wpk, gwp, gxt, a);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

// stage bits: 1 the gather kernel, 2 the streaming kernel + gx (fc_backward_gather / fc_backward_stream time them apart), 4 without the gx
// kernel (fc_backward_all with module parameters: the finishing launch forms gx)
int backward_stream_impl(const float* x, const float* gy, const float* rec, const fc_csr* g, const float* wpk, float* gx, void* ws,
                         size_t ws_bytes, const fc_dims* d, hipStream_t stream, int stages) {
    const StreamPlan p = plan_stream(d, halves_of(d), true);
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hrec_bytes + p.gwp_bytes + p.gxt_bytes + p.wst_bytes) return FC_ERR_WORKSPACE;
    StreamArgs a = make_stream_args(d, p);
    char* hrec = static_cast<char*>(ws);
    a.wst = reinterpret_cast<uint32_t*>(hrec + p.hrec_bytes + p.gwp_bytes + p.gxt_bytes);      // (16-byte aligned: the three sizes before it are)
    float2* gwp = reinterpret_cast<float2*>(hrec + p.hrec_bytes);
    float2* gxt = reinterpret_cast<float2*>(hrec + p.hrec_bytes + p.gwp_bytes);
    const float2* x2 = reinterpret_cast<const float2*>(x);
    int rc = FC_ERR_UNSUPPORTED;
    if (stages & 1) {
        const float2* gy2 = reinterpret_cast<const float2*>(gy);
#define FC_GATHER_CASE(RR, BB) if (d->R == RR && d->B == BB) rc = launch_gather<RR, BB>(gy2, rec, g, wpk, hrec, a, p, stream);
        FC_GATHER_CASE(2, 1) FC_GATHER_CASE(4, 1) FC_GATHER_CASE(6, 1) FC_GATHER_CASE(8, 1)
        FC_GATHER_CASE(2, 2) FC_GATHER_CASE(4, 2) FC_GATHER_CASE(6, 2) FC_GATHER_CASE(8, 2)
        FC_GATHER_CASE(2, 3) FC_GATHER_CASE(4, 3) FC_GATHER_CASE(6, 3) FC_GATHER_CASE(8, 3)
#undef FC_GATHER_CASE
        if (rc != FC_OK) return rc;
    }
    if (!(stages & 2)) return FC_OK;
    if (p.T == 6 && p.KPS == 288 && d->I == 48) rc = launch_stream<6, 288, 48>(x2, hrec, wpk, gwp, gxt, a, p, stream);      // the reference's default layer
    else if (p.T == 6 && p.KPS == 192 && d->I == 64) rc = launch_stream<6, 192, 64>(x2, hrec, wpk, gwp, gxt, a, p, stream);  // config 5's layer: a half of 64 x 6 entries
    else if (p.T <= 2) rc = launch_stream<2>(x2, hrec, wpk, gwp, gxt, a, p, stream);
    else if (p.T <= 4) rc = launch_stream<4>(x2, hrec, wpk, gwp, gxt, a, p, stream);
    else rc = launch_stream<6>(x2, hrec, wpk, gwp, gxt, a, p, stream);
    if (rc != FC_OK || (stages & 4)) return rc;         // (4: gx is left to the launch that finishes the pass)
    const size_t count = (size_t)d->N * d->I;
    const dim3 grid((unsigned)((count + 255) / 256));
    float2* gx2 = reinterpret_cast<float2*>(gx);
    if (d->B == 1) hipLaunchKernelGGL(fc_backward_gx_kernel<1>, grid, dim3(256), 0, stream, x2, gxt, gx2, count, p.KS);
    else if (d->B == 2) hipLaunchKernelGGL(fc_backward_gx_kernel<2>, grid, dim3(256), 0, stream, x2, gxt, gx2, count, p.KS);
    else hipLaunchKernelGGL(fc_backward_gx_kernel<3>, grid, dim3(256), 0, stream, x2, gxt, gx2, count, p.KS);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

bool backward_streams(const fc_dims* d, bool factored) { return plan_stream(d, halves_of(d), factored).ok; }

bool backward_fits(const fc_dims* d) { return plan_backward(d, halves_of(d)).ok_factored; }

size_t backward_workspace_bytes(const fc_dims* d) {
    const BwdPlan p = plan_backward(d, halves_of(d));
    const size_t pair = p.hdump_bytes + p.gwp_bytes + p.gxp_bytes + 256;
    const StreamPlan sp = plan_stream(d, halves_of(d), true);          // (records or not is the launch's choice: room for either)
    const size_t stream = sp.ok ? sp.hrec_bytes + sp.gwp_bytes + sp.gxt_bytes + sp.wst_bytes + 256 : 0;
    return pair > stream ? pair : stream;
}

template <int T>
static int launch_backward_filter(const float2* x, const float* hdump, float2* gwp, const BwdArgs& a, const BwdPlan& p,
                                  const fc_dims* d, hipStream_t stream) {
    static const bool half2 = !(dev_env("FC_FILTER2") && atoi(dev_env("FC_FILTER2")) == 0);      // FC_FILTER2=0: the LDS-staged kernel (development)
    static bool ok_half2[kMaxDevices] = {}, ok_half[kMaxDevices] = {}, ok_f32[kMaxDevices] = {};      // per T (this function is a template)
    if (p.fhalf && half2) {
        const size_t lds2 = (size_t)(2 * kTile * filter_image_stride(p.KP) + 2 * 6 * p.IP * kXbStride + 8) * sizeof(_Float16) + 16;
        static bool ok_half2h[kMaxDevices] = {};
        if (a.dump_halves) {
            if (!allow_full_lds(reinterpret_cast<const void*>(fc_backward_filter_half2_kernel<T, true>), lds2, ok_half2h)) return FC_ERR_LAUNCH;
            hipLaunchKernelGGL((fc_backward_filter_half2_kernel<T, true>), dim3(p.P, p.F), dim3(kThreads), lds2, stream, x, hdump, gwp, a, p.F,
                               d->B, d->R * d->O);
        } else {
            if (!allow_full_lds(reinterpret_cast<const void*>(fc_backward_filter_half2_kernel<T, false>), lds2, ok_half2)) return FC_ERR_LAUNCH;
            hipLaunchKernelGGL((fc_backward_filter_half2_kernel<T, false>), dim3(p.P, p.F), dim3(kThreads), lds2, stream, x, hdump, gwp, a, p.F,
                               d->B, d->R * d->O);
        }
    } else if (p.fhalf) {
        if (!allow_full_lds(reinterpret_cast<const void*>(fc_backward_filter_half_kernel<T>), p.lds_filter, ok_half)) return FC_ERR_LAUNCH;
        hipLaunchKernelGGL(fc_backward_filter_half_kernel<T>, dim3(p.P, p.F), dim3(kThreads), p.lds_filter, stream, x, hdump, gwp, a,
                           p.F, d->B, d->R * d->O);
    } else {
        if (!allow_full_lds(reinterpret_cast<const void*>(fc_backward_filter_kernel<T>), p.lds_filter, ok_f32)) return FC_ERR_LAUNCH;
        hipLaunchKernelGGL(fc_backward_filter_kernel<T>, dim3(p.P, p.F), dim3(kThreads), p.lds_filter, stream, x, hdump, gwp, a, p.F,
                           d->B);
    }
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int backward_filter_impl(const float* x, void* ws, size_t ws_bytes, const fc_dims* d, hipStream_t stream, bool factored) {
    if (backward_streams(d, factored)) return FC_OK;        // (the streaming kernel behind the gather has left the partials already)
    const BwdPlan p = plan_backward(d, halves_of(d));
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const BwdArgs a = make_args(d, p);
    const float* hdump = reinterpret_cast<const float*>(ws);
    float2* gwp = reinterpret_cast<float2*>(static_cast<char*>(ws) + p.hdump_bytes);
    const float2* x2 = reinterpret_cast<const float2*>(x);
    int need = (p.ngw + kWaves - 1) / kWaves;
    if (p.fhalf) {      // column-grouped assignment: the wavefronts of the smallest column group bound the slot count
        const int per_col = kWaves / p.g.NMT, row_tiles = p.KP / 16;
        need = (row_tiles + per_col - 1) / per_col;
    }
    if (need <= 2) return launch_backward_filter<2>(x2, hdump, gwp, a, p, d, stream);
    if (need <= 4) return launch_backward_filter<4>(x2, hdump, gwp, a, p, d, stream);
    if (need <= 6) return launch_backward_filter<6>(x2, hdump, gwp, a, p, d, stream);      // 64 channels x 6 rings: 24 row tiles over 4 wavefronts per column
    if (need <= kMaxGwTiles) return launch_backward_filter<kMaxGwTiles>(x2, hdump, gwp, a, p, d, stream);
    return FC_ERR_UNSUPPORTED;
}

// Which kernels a backward pass with these dims launches (fc_describe_kernels).
void describe_backward(const fc_dims* d, int records, char* buf, size_t n) {
    {
        const StreamPlan sp = plan_stream(d, halves_of(d), records != 0);
        if (sp.ok) {
            snprintf(buf, n, "fc_backward_gather_kernel<records,split-f16> tiles=%d; fc_backward_stream_kernel (H once for gxt and gW: %d gxt + %d gW "
                     "wavefronts, W_f in registers, records by LDS-DMA) grid=%dx%d; fc_backward_gx_kernel; module parameters: "
                     "fc_backward_finish_params (sum of the partials + parameter chain) in one launch; explicit filter: fc_backward_finish; cus=%d",
                     sp.ntiles, sp.G, sp.NW, sp.P, sp.FS, num_cus());
            return;
        }
    }
    const BwdPlan p = plan_backward(d, halves_of(d));
    const char* mode = halves_of(d) == 2 ? "split-f16" : halves_of(d) == 1 ? "f16" : "f32";
    static const bool staged = [] { const char* e = dev_env("FC_FILTER2"); return e && atoi(e) == 0; }();
    static const bool split_finish = [] { const char* e = dev_env("FC_SPLIT_FINISH"); return e && atoi(e) != 0; }();
    // what ends the pass follows the same switches the launches use (fc_api.hip: fc_backward_all / fc_backward_finish_params)
    const int gx_parts = (1 << p.parts_log2) * (p.gsplit ? 2 : 1);
    char fin[320];
    if (split_finish)
        snprintf(fin, sizeof(fin), "%sfc_backward_finish + fc_filter_param_grads (two launches: the split-finish development switch)",
                 gx_parts > 1 ? "fc_sum_parts_kernel (the partial gx arrays), " : "");
    else
        snprintf(fin, sizeof(fin), "module parameters: fc_backward_finish_params (sum of the partials + parameter chain%s, + a fused modReLU's "
                 "bias-gradient sum when given) in one launch; explicit filter: %sfc_backward_finish",
                 gx_parts > 1 ? " + sum of the partial gx arrays" : "", gx_parts > 1 ? "fc_sum_parts_kernel, " : "");
    snprintf(buf, n, "fc_backward_data_kernel<%s,%s> tiles=%d parts=%d%s; %s; %s; cus=%d", records ? "records" : "dense rows", mode, p.ntiles,
             1 << p.parts_log2, p.gsplit ? " (the two frequency groups of a tile as separate work items)" : "",
             p.fhalf ? (staged ? "fc_backward_filter_half_kernel (LDS-staged slabs)" : "fc_backward_filter_half2_kernel (register-fed rows)")
                     : "fc_backward_filter_kernel (fp32 MFMA)",
             fin, num_cus());
}

// Fixed-order sum of the per-workgroup filter-gradient partials left in the workspace.
int backward_finish_impl(float* gw_eff, void* ws, size_t ws_bytes, const fc_dims* d, hipStream_t stream, bool factored) {
    {
        const StreamPlan sp = plan_stream(d, halves_of(d), factored);
        if (sp.ok) {        // partials [P][F][k = o*R + r][IP]
            if (!ws || ws_bytes < sp.hrec_bytes + sp.gwp_bytes) return FC_ERR_WORKSPACE;
            const float2* gwp = reinterpret_cast<const float2*>(static_cast<char*>(ws) + sp.hrec_bytes);
            const int total = sp.F * d->R * d->O * d->I;
            hipLaunchKernelGGL(fc_reduce_gw_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, gwp, reinterpret_cast<float2*>(gw_eff),
                               sp.P, sp.F, d->R, d->O, d->I, sp.KP, sp.IP, 2);
            return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
        }
    }
    const BwdPlan p = plan_backward(d, halves_of(d));
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const float2* gwp = reinterpret_cast<const float2*>(static_cast<char*>(ws) + p.hdump_bytes);
    const int total = p.F * d->R * d->O * d->I;
    hipLaunchKernelGGL(fc_reduce_gw_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, gwp,
                       reinterpret_cast<float2*>(gw_eff), p.P, p.F, d->R, d->O, d->I, p.KP, p.IP, p.gd.split != 0 ? 1 : 0);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

int backward_finish_params_impl(float* gw_eff, void* ws, size_t ws_bytes, const fc_dims* d, const fc_filter_params* fp, hipStream_t stream,
                                int o0, int i0, int Ifull, float* gx_deferred, bool factored, const float* x_for_gx) {
    {
        const StreamPlan sp = plan_stream(d, halves_of(d), factored);
        if (sp.ok) {        // partial (p, f, k = o*R + r, i) at ((p*F + f)*KP + k)*IP + i
            if (!ws || ws_bytes < sp.hrec_bytes + sp.gwp_bytes + sp.gxt_bytes) return FC_ERR_WORKSPACE;
            const float* gwp = reinterpret_cast<const float*>(static_cast<char*>(ws) + sp.hrec_bytes);
            // gx_deferred (fc_backward_all): the gx kernel was left out -- gx is formed from the gxt slices by extra workgroups of this launch
            const bool ride = gx_deferred && x_for_gx;
            const float* gxt = reinterpret_cast<const float*>(static_cast<char*>(ws) + sp.hrec_bytes + sp.gwp_bytes);
            return reduce_param_grads_impl(gwp, (size_t)sp.F * sp.KP * sp.IP, (size_t)sp.IP, (size_t)sp.KP * sp.IP, (size_t)d->R * sp.IP, false, sp.P,
                                           gw_eff, fp->zonal, fp->spherical, fp->phase, fp->ftype, fp->g_zonal, fp->g_spherical, fp->g_phase, d,
                                           stream, o0, i0, Ifull, fp->bias_partials, fp->bias_nparts, fp->g_bias, ride ? gxt : nullptr,
                                           ride ? gx_deferred : nullptr, (size_t)d->N * d->I, 0, ride ? -(d->B + 4 * (sp.KS - 1)) : 0, ride ? x_for_gx : nullptr);
        }
    }
    const BwdPlan p = plan_backward(d, halves_of(d));
    if (!p.ok) return FC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < p.hdump_bytes + p.gwp_bytes) return FC_ERR_WORKSPACE;
    const float* gwp = reinterpret_cast<const float*>(static_cast<char*>(ws) + p.hdump_bytes);
    // the data kernel's partial gx arrays (backward_data_impl with defer_gx_sum), added here
    const int gx_nparts = gx_deferred ? (1 << p.parts_log2) << p.gsplit : 1;
    const float* gxp = gx_nparts > 1 ? reinterpret_cast<const float*>(static_cast<char*>(ws) + p.hdump_bytes + p.gwp_bytes) : nullptr;
    // partial (p, f, k = r*O + o, i) at ((p*F + f)*KP + k)*IP + i
    return reduce_param_grads_impl(gwp, (size_t)p.F * p.KP * p.IP, (size_t)d->O * p.IP, (size_t)p.KP * p.IP, (size_t)p.IP, p.gd.split != 0, p.P, gw_eff,
                                   fp->zonal, fp->spherical, fp->phase, fp->ftype, fp->g_zonal, fp->g_spherical, fp->g_phase, d, stream, o0,
                                   i0, Ifull, fp->bias_partials, fp->bias_nparts, fp->g_bias, gxp, gx_deferred, (size_t)d->N * d->I,
                                   p.gx_part_stride, gx_nparts);
}

#define FC_BWD_DATA_ARGS const float*, const float*, const float*, const fc_csr*, const float*, float*, void*, size_t, const fc_dims*, bool, hipStream_t, bool
template int backward_data_impl_mode<false>(FC_BWD_DATA_ARGS);
extern template int backward_data_impl_mode<true>(FC_BWD_DATA_ARGS);

int backward_data_impl(const float* x, const float* gy, const float* sten, const fc_csr* g, const float* wpk, float* gx,
                       void* ws, size_t ws_bytes, const fc_dims* d, bool factored, hipStream_t stream, bool defer_gx_sum) {
    // large meshes: gather kernel + the kernel that streams H once for gxt and gW + gx -- after this call gx is complete AND the
    // filter-gradient partials are in the workspace (backward_filter_impl has nothing left to do)
    if (backward_streams(d, factored)) return backward_stream_impl(x, gy, sten, g, wpk, gx, ws, ws_bytes, d, stream, defer_gx_sum ? 7 : 3);
    return halves_of(d) ? backward_data_impl_mode<true>(x, gy, sten, g, wpk, gx, ws, ws_bytes, d, factored, stream, defer_gx_sum)
                        : backward_data_impl_mode<false>(x, gy, sten, g, wpk, gx, ws, ws_bytes, d, factored, stream, defer_gx_sum);
}

}  // namespace fc
