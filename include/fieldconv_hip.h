/* fieldconv_hip.h -- C ABI of libfieldconv_hip.so (gfx950 / MI355X).
 *
 * The reference (twmitchel/FieldConv) has no native interface for this path: its hot loop is a
 * composite of stock torch ops plus torch_scatter.scatter_add inside Python modules.  Each entry
 * point below replaces the body of one reference Python method; a maintainer binds it with
 * ctypes / a torch custom op exactly as fieldconv_amd/_lib.py does.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch); nothing is allocated,
 *     freed or retained by the library; workspaces are caller-provided
 *   - complex tensors are interleaved (re, im) fp32 pairs == torch.complex64 storage
 *   - indices are int32; `stream` is a hipStream_t passed as void*
 *   - return value: 0 on success, a negative fc_status otherwise; no exceptions, no globals,
 *     re-entrant; work is enqueued on `stream` and NOT synchronised
 */
#ifndef FIELDCONV_HIP_H
#define FIELDCONV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum fc_status {
    FC_OK = 0,
    FC_ERR_BAD_ARGUMENT = -1,   /* null pointer, non-positive size, inconsistent dims */
    FC_ERR_UNSUPPORTED = -2,    /* (n_rings, band_limit, channels) outside the compiled set */
    FC_ERR_LAUNCH = -3,         /* hipLaunchKernel reported an error */
    FC_ERR_WORKSPACE = -4       /* workspace pointer missing or too small */
} fc_status;

/* Problem dimensions of one FieldConv call (reference nn/field_conv.py:104-123). */
typedef struct fc_dims {
    int32_t N;   /* vertices                                   */
    int32_t E;   /* support edges                              */
    int32_t I;   /* in_channels                                */
    int32_t O;   /* out_channels                               */
    int32_t R;   /* n_rings                                    */
    int32_t B;   /* band_limit; F = 2B+1 angular frequencies   */
    int32_t mode; /* fc_mfma_mode: arithmetic of the contractions (0 = FC_MFMA_SPLIT_F16, the default) -- see below.  Part of the dims
                   * of EVERY call: size queries, packing calls and launches of one convolution carry the same value; two convolutions
                   * in one process may differ (the library keeps no arithmetic state) */
} fc_dims;

/* Edge list grouped by one endpoint (CSR).  Slot s in [rowptr[v], rowptr[v+1]) is one edge
 * incident to vertex v; nbr[s] is the OTHER endpoint.  Grouped by target (col 1 of supp_edges)
 * for the forward pass, by source (col 0) for the backward pass.  The stencil passed alongside a
 * CSR is stored in that CSR's SLOT order (the kernels stream it): see fc_forward / fc_backward. */
typedef struct fc_csr {
    const int32_t* rowptr;   /* N+1 */
    const int32_t* nbr;      /* E   (dense entry points; may be NULL for the factored ones) */
    const int32_t* runs;     /* factored entry points only, else NULL: N x 8 ints; runs[8v + q] = first slot of
                              * vertex v, relative to rowptr[v], whose ring index is >= q (so the slots of ring q
                              * are [runs[8v+q], runs[8v+q+1]) and runs[8v] = 0) */
} fc_csr;

int fc_abi_version(void);
/* Arithmetic of the three contractions (forward, input gradient, filter gradient), carried PER CALL in fc_dims::mode / fc_mesh::mode --
 * no process-wide setter, no environment variable: FC_MFMA_SPLIT_F16 (default) carries every operand as two f16 halves with
 * power-of-two row scales on v_mfma_f32_16x16x32_f16, fp32 accumulation (fp32-grade: 2^-22 of the row maximum); FC_MFMA_F32 runs
 * v_mfma_f32_16x16x4_f32 on fp32 operands throughout; FC_MFMA_F16 drops the low halves (reduced precision, ~3e-4).  Packed filter
 * images, workspaces and kernels all follow the mode: a packing call and the launches that use its images carry the same one. */
typedef enum fc_mfma_mode { FC_MFMA_SPLIT_F16 = 0, FC_MFMA_F32 = 1, FC_MFMA_F16 = 2 } fc_mfma_mode;
/* 1 when this library was compiled with -DFC_DEV_SWITCHES (libfieldconv_hip_dev.so: the FC_* development variables of
 * fieldconv_amd/_env.py select older kernel families, skip phases, write stamps); the product library returns 0 and reads no variable. */
int fc_dev_switches(void);
/* Development only (tools/stamps.py): a device buffer of 16 x 256 uint64 that the record-driven kernels fill with in-kernel
 * time stamps of workgroup 0 (label << 56 | s_memtime), or NULL to switch the stamps off (the default).  Process-wide. */
void fc_debug_stamp_buffer(void* device_buffer);
const char* fc_status_string(int status);

/* 1 if the compiled kernels cover these dims, 0 otherwise (then every call returns FC_ERR_UNSUPPORTED for them):
 * (n_rings, band_limit) among the compiled shapes, at most 64 channels, and slab + partial sums + record ring within
 * the CU's 160 KB of LDS in the current MFMA mode (8 rings with more than 56 channels are not, in the default mode).
 * The operator is linear in the input channels and independent across output channels: wider layers run as channel blocks
 * (fc_forward_wide / fc_backward_wide below). */
int fc_supported(const fc_dims* dims);
/* Which kernels a forward + backward pass with these dims launches in this process, as one line of text (kernel family,
 * record kind, MFMA mode, tile counts): the library picks them from the dims, the device's CU count, the MFMA mode and -- in the
 * development build only -- its switches, and a benchmark line should say what it timed.  kind: 0 dense rows, 1 factored
 * records, 2 geometric records in the forward pass (factored ones in the backward pass).  No GPU work. */
int fc_describe_kernels(const fc_dims* dims, int32_t kind, char* buffer, size_t buffer_bytes);

/* ---- filter packing -------------------------------------------------------------------- *
 * W_eff (O,I,R,F) complex64 is what reference nn/field_conv.py:10-33 (weightContrib*) builds
 * from (zonal, spherical, phase); the 1/(2B+1) of :14,:25,:33 is folded in here.
 * The packed images are opaque to the caller: allocate fc_packed_filter_floats_{fwd,bwd}() floats
 * each and hand them to the convolution calls.  Default ("split") layout, one image per contraction:
 *   OP inverse row scales (floats), then F x {re_hi, re_lo, im_hi, im_lo} x KP/32 k blocks x OP x 32 halves,
 *   forward rows o, k = r*ceil8(I) + i;  backward rows i, k = r*ceil8(O) + o, conjugated;
 *   OP = ceil16(rows), KP = ceil32(R * ceil8(channels)).
 * In FC_MFMA_F32 mode (fc_dims::mode): F x {re,im} x OP x ceil16(R*channels) floats.
 * `records`: which family of convolution entry points the images are for -- 0: the dense-stencil ones (fc_forward,
 * fc_backward_data), 1: the record-driven ones (fc_forward_factored, fc_forward_geometric, fc_backward_data_factored).
 * In the default mode the record-driven forward image is RING-major: OP inverse row scales, then
 * R x {re_hi, re_lo, im_hi, im_lo} x KP/32 k blocks x OP x 32 halves with k = f*ceil8(I) + i, KP = ceil32(F * ceil8(I))
 * (csrc/fc_forward_ring.hpp) whenever the forward launch with these dims takes the ring-major kernel (meshes of more than
 * 256 tiles of 16 vertices): the layout follows dims->N, so an image is packed with the dims of the launch it is for
 * (a forward pass launched in two row ranges packs one image per range).  Sizes differ between the two families, ask with
 * the same `records`.  wpk_bwd may be NULL: only the forward image is written. */
/* `records` for every call below that takes one: 0 for the dense-stencil entry points, 1 for the record-driven ones
 * (fc_records_flags(dims, record_driven) returns exactly that; kept so that a binding never hard-codes the value).
 * Either image pointer of the packing calls may be NULL (not both): only the other image is written. */
int32_t fc_records_flags(const fc_dims* dims, int32_t record_driven);
size_t fc_packed_filter_floats_fwd(const fc_dims* dims, int32_t records);
size_t fc_packed_filter_floats_bwd(const fc_dims* dims, int32_t records);
int fc_pack_filter(const float* w_eff, float* wpk_fwd, float* wpk_bwd, const fc_dims* dims, int32_t records, void* stream);
/* Same, assembling W_eff on the fly from the module parameters exactly as reference
 * nn/field_conv.py:10-33 does (zonal (O,I,R[,2]), spherical (O,I,R,B|2B,2), phase (O,I,B+1), all fp32
 * contiguous; ftype 0/1/2 as in :53-59; phase is read for ftype 1 only). */
int fc_pack_filter_params(const float* zonal, const float* spherical, const float* phase, int32_t ftype,
                          float* wpk_fwd, float* wpk_bwd, const fc_dims* dims, int32_t records, void* stream);
/* Autograd twin of the assembly: dL/dW_eff (O,I,R,F) c64 -> gradients shaped like the parameters
 * (overwritten; g_phase written for ftype 1 only). */
int fc_filter_param_grads(const float* gw_eff, const float* zonal, const float* spherical, const float* phase,
                          int32_t ftype, float* g_zonal, float* g_spherical, float* g_phase,
                          const fc_dims* dims, void* stream);

/* Optional epilogue of the forward convolutions (reference nn/fc_resnet_block.py:84-88): what follows a
 * FieldConv inside an FCResNetBlock / ECHOBlock is applied to the output tile before it leaves the kernel.
 *   addend        (N,O) c64 or NULL   added to the convolution's output (the block's TangentLin residual)
 *   modrelu_bias  O floats or NULL    TangentNonLin (reference nn/tangent_nonlin.py:24-35) of (conv + addend)
 *   activated     (N,O) c64           receives the activated values; required with modrelu_bias
 * `y` always receives conv + addend (the pre-activation: the backward pass of the modReLU needs it).  NULL epilogue = none. */
typedef struct fc_epilogue {
    const float* addend;
    const float* modrelu_bias;
    float* activated;
} fc_epilogue;

/* ---- FieldConv.forward, reference nn/field_conv.py:128-137 ------------------------------- *
 * y[n,o] = 1/F sum_{e: dst_e=n} sum_{i,r,f} x[src_e,i] e^{-i(f-B)phi[src_e,i]} S[e,r,f] W_eff[o,i,r,f]
 * x (N,I) c64; by_target: CSR grouped by target; sten_t (E,R,F) c64 = supp_sten rows permuted into
 * by_target slot order; y (N,O) c64 (overwritten). */
int fc_forward(const float* x, const float* sten_t, const fc_csr* by_target, const float* wpk_fwd,
               float* y, const fc_dims* dims, const fc_epilogue* epilogue, void* stream);

/* ---- factored stencil fast path ---------------------------------------------------------- *
 * FCPrecomp's stencil (reference transforms/fc_precomp.py:24-25,95) is rank-1 and 2-sparse in the
 * ring index: supp_sten[e,r,f] = w[e,r] * ph[e,f] with w[e,q], w[e,q+1] the only non-zeros.  When
 * the caller has verified that structure it may pass, instead of the dense rows, one record of
 * fc_factored_record_floats(B) floats per edge (in by_target slot order and, inside every target's
 * slot range, sorted by q; followed by at least 1 KiB of readable padding):  [0] q as int32 bits, [1] w[q], [2] w[q+1],
 * [3] the slot's other endpoint (= by_target->nbr[slot]) as int32 bits, [4+2f], [5+2f] = Re, Im ph[f].  Same result as fc_forward up to fp32 rounding.
 * workspace (optional, may be NULL / 0): fc_forward_workspace_bytes(dims) bytes of scratch.  On meshes whose N/16 vertex
 * tiles cannot occupy the 256 CUs (the reference's segmentation meshes: ~1k vertices with ~128 neighbours) it lets up
 * to 8 workgroups share a tile, each taking a share of every target's edges; their partial outputs are added in a
 * fixed order.  fc_forward_workspace_bytes returns 0 when the split does not apply. */
int fc_factored_record_floats(int32_t band_limit);
size_t fc_forward_workspace_bytes(const fc_dims* dims);
int fc_forward_factored(const float* x, const float* rec_t, const fc_csr* by_target, const float* wpk_fwd,
                        float* y, void* workspace, size_t workspace_bytes, const fc_dims* dims, const fc_epilogue* epilogue,
                        void* stream);

/* ---- geometric-phase records (forward only) ---------------------------------------------- *
 * FCPrecomp's phases are geometric in the frequency: ph[e,f] = c[e] * g[e]^(f-B) with |g| = 1
 * (fSten = exp(i m theta) times the per-edge weight, reference transforms/fc_precomp.py:88-95).  When the
 * caller has verified that as well, the forward pass takes records of fc_geometric_record_floats() = 8
 * floats per edge, same order and padding as above:  [0] q bits, [1] w[q], [2] w[q+1], [3] other endpoint
 * bits, [4],[5] = Re, Im c, [6],[7] = Re, Im g.  Same result as fc_forward up to fp32 rounding. */
int fc_geometric_record_floats(void);
int fc_forward_geometric(const float* x, const float* geo_t, const fc_csr* by_target, const float* wpk_fwd,
                         float* y, void* workspace, size_t workspace_bytes, const fc_dims* dims, const fc_epilogue* epilogue,
                         void* stream);

/* ---- autograd of the above (the reference relies on torch autograd through :128-137) ----- *
 * Three calls on the same stream, sharing `workspace` (fc_backward_workspace_bytes(dims) bytes,
 * 256-byte aligned, untouched in between):
 *   fc_backward_data[_factored]  one kernel: source-centric gather of H = sum gy conj(S) over the
 *                                out-edges (by_source: CSR grouped by source; sten_s / rec_s in its
 *                                slot order, records sorted by ring inside a source and carrying the
 *                                target in [3]), MFMA contraction with the conjugated filter, writes
 *                                gx (N,I) c64 and leaves H in the workspace
 *   fc_backward_filter           one kernel: filter-gradient partials from the stored H and x
 *   fc_backward_finish           fixed-order sum of the partials into gw_eff (O,I,R,F) c64 */
size_t fc_backward_workspace_bytes(const fc_dims* dims, int32_t records);
int fc_backward_data(const float* x, const float* gy, const float* sten_s, const fc_csr* by_source,
                     const float* wpk_bwd, float* gx, void* workspace, size_t workspace_bytes,
                     const fc_dims* dims, void* stream);
int fc_backward_data_factored(const float* x, const float* gy, const float* rec_s, const fc_csr* by_source,
                              const float* wpk_bwd, float* gx, void* workspace, size_t workspace_bytes,
                              const fc_dims* dims, int32_t records, void* stream);
int fc_backward_filter(const float* x, void* workspace, size_t workspace_bytes, const fc_dims* dims, int32_t records, void* stream);
/* Large meshes on records in the default arithmetic mode (fc_backward_streams(dims, records) != 0) run the H-STREAMING arrangement
 * behind the same three calls: fc_backward_data_factored = a gather kernel (H of every source vertex to the workspace, nothing else) +
 * ONE kernel that streams H once for both products (gxt = H conj(W), gW = H^T conj(xt)) + a small kernel that forms gx; gx is complete
 * and the filter-gradient partials are in the workspace when its work is done, and fc_backward_filter has nothing left to launch.
 * fc_backward_gather / fc_backward_stream are the two halves of that fc_backward_data_factored call as entry points of their own
 * (per-kernel timing; FC_ERR_UNSUPPORTED where the arrangement does not apply).  The gather launch also leaves, in the workspace, a
 * copy of the packed filter in the order the streaming launch's wavefronts load their fragments: fc_backward_stream reads what the
 * fc_backward_gather call before it wrote for the same wpk_bwd.  Autograd of reference nn/field_conv.py:21,130,134. */
int32_t fc_backward_streams(const fc_dims* dims, int32_t records);
int fc_backward_gather(const float* gy, const float* rec_s, const fc_csr* by_source, const float* wpk_bwd, void* workspace,
                       size_t workspace_bytes, const fc_dims* dims, void* stream);
int fc_backward_stream(const float* x, const float* wpk_bwd, float* gx, void* workspace, size_t workspace_bytes, const fc_dims* dims,
                       void* stream);
int fc_backward_finish(float* gw_eff, void* workspace, size_t workspace_bytes, const fc_dims* dims, int32_t records, void* stream);

/* ---- one call per pass: the same kernels, enqueued by one entry point (a binding that pays microseconds per foreign call --
 * ctypes -- spends more time on six calls per convolution than the GPU needs for a small mesh) ---------------------------- *
 * fc_filter_params: the module parameters as fc_pack_filter_params takes them and, for the backward pass, where their
 * gradients go (g_* are ignored by fc_forward_params; g_phase / phase may be NULL unless ftype == 1).
 * fc_forward_params = fc_pack_filter_params (wpk_bwd may be NULL; `records` as above) + fc_forward (kind 0) / fc_forward_factored (1) /
 *                     fc_forward_geometric (2), same arguments;
 * fc_backward_all   = fc_backward_data (records 0) or fc_backward_data_factored (1) + fc_backward_filter +
 *                     fc_backward_finish into gw_eff + fc_filter_param_grads when params is not NULL (then the pass
 *                     ends in fc_backward_finish_params' single launch, which also adds the data kernel's partial gx arrays when
 *                     tiles were shared: gx is complete when the call's work is, not between its kernels).  gw_eff is required
 *                     without params and OPTIONAL with them: NULL = only the parameter gradients are wanted (a module's backward
 *                     pass; the (O,I,R,F) tensor is then never written). */
typedef struct fc_filter_params {
    const float* zonal;
    const float* spherical;
    const float* phase;
    int32_t ftype;
    float* g_zonal;
    float* g_spherical;
    float* g_phase;
    /* optional rider of the launch that finishes the backward pass (fc_backward_finish_params / fc_backward_all): the fixed-order sum of
     * the modReLU's bias-gradient partials -- bias_partials (bias_nparts, O) floats as fc_tangent_nonlin_backward_partial left them --
     * into g_bias (O), instead of a launch of its own.  NULL / 0: nothing. */
    const float* bias_partials;
    int32_t bias_nparts;
    float* g_bias;
} fc_filter_params;
int fc_forward_params(const float* x, const float* sten_or_records, const fc_csr* by_target, int32_t kind,
                      const fc_filter_params* params, float* wpk_fwd, float* wpk_bwd, float* y, void* workspace,
                      size_t workspace_bytes, const fc_dims* dims, int32_t records, const fc_epilogue* epilogue, void* stream);
int fc_backward_all(const float* x, const float* gy, const float* sten_or_rec_s, const fc_csr* by_source, int32_t records,
                    const float* wpk_bwd, float* gx, float* gw_eff, const fc_filter_params* params, void* workspace,
                    size_t workspace_bytes, const fc_dims* dims, void* stream);
/* fc_backward_finish + fc_filter_param_grads in ONE launch: every workgroup sums the partials of its
 * (output channel, 16 input channels) block in the fixed order, writes gw_eff (unless NULL) and pulls the block back to the parameters.
 * The partials are summed in four consecutive groups whose sums are then added in order: deterministic, but rounded
 * differently from fc_backward_finish's single chain.  fc_backward_all uses it when it is given params. */
int fc_backward_finish_params(float* gw_eff, void* workspace, size_t workspace_bytes, const fc_dims* dims, int32_t records,
                              const fc_filter_params* params, void* stream);

/* ---- layers wider than 64 channels, natively (reference nn/field_conv.py:62 takes any width) ----------------------------------- *
 * The kernels above take at most 64 input and 64 output channels (one channel per lane in their gather phases).  The operator is
 * linear in the input channels and independent across output channels, so a wider layer is nob x nib launches of the same
 * kernels on channel blocks of `block` channels (a multiple of 8, <= 64, with fc_supported(block x block) true); these two entry
 * points enqueue a whole pass from one call -- strided 2-D copies cut x / gy into contiguous blocks, the filter images are packed
 * straight from the (o0, i0) block of the full parameter tensors (or of an explicit W_eff), the sum over input blocks rides in the
 * convolution's residual epilogue, the input gradient's sum over output blocks is a fixed-order sum of partials, and a block's
 * parameter gradients land in their block of the full gradient tensors (params->g_*), or in gw_eff (O,I,R,F) for an explicit filter.
 * Exactly one of `params` / `w_eff` is given.  dims carries the FULL widths I, O; y (N,O), gx (N,I) complex64, overwritten.
 * workspace: fc_wide_workspace_bytes(dims, block, records, backward) bytes. */
size_t fc_wide_workspace_bytes(const fc_dims* dims, int32_t block, int32_t records, int32_t backward);
int fc_forward_wide(const float* x, const float* sten_or_records, const fc_csr* by_target, int32_t kind, const fc_filter_params* params,
                    const float* w_eff, float* y, void* workspace, size_t workspace_bytes, const fc_dims* dims, int32_t records,
                    int32_t block, void* stream);
int fc_backward_wide(const float* x, const float* gy, const float* sten_or_rec_s, const fc_csr* by_source, int32_t records,
                     const fc_filter_params* params, const float* w_eff, float* gw_eff, float* gx, void* workspace,
                     size_t workspace_bytes, const fc_dims* dims, int32_t block, void* stream);

/* ---- any (n_rings, band_limit): the run-time path for shapes outside the compiled set (reference nn/field_conv.py:62-98
 * takes any) ---- *
 * fc_shape_compiled: 1 for n_rings 2..8 x band_limit 1..3 (the specialised kernels above), else 0.  For the others -- and for
 * complex128 features of ANY shape (the reference's modules run under .double()) -- the operator is evaluated in the reference's
 * own two steps, with dense stencil rows in slot order, in the tensors' own precision (dtype):
 *   fc_generic_gather   contrib (n_targets, I, R, F) c64 = the per-target response of :128-134 (by_target CSR with nbr; sten_t
 *                       (E,R,F) complex in its slot order); the caller contracts it with W_eff and forms the two backward
 *                       products with fc_cgemm (below);
 *   fc_generic_scatter  gx (N, I) c64 from g_contrib (n_targets, I, R, F) over the out-edges (by_source CSR with nbr = targets,
 *                       sten_s in its slot order), including the chain rule through the rotation e^{-i m angle(x)}.
 * Any channel count: a workgroup takes as many input channels as its R * F accumulators per channel fit in the CU's 160 KB of LDS. */
typedef enum fc_dtype { FC_F32 = 0, FC_F64 = 1 } fc_dtype;       /* real scalar type of a complex tensor: complex64 / complex128 */
int fc_shape_compiled(int32_t n_rings, int32_t band_limit);
int fc_generic_gather(const void* x, const void* sten_t, const fc_csr* by_target, void* contrib, int32_t n_targets, int32_t I,
                      int32_t R, int32_t B, int32_t dtype, void* stream);
int fc_generic_scatter(const void* x, const void* g_contrib, const void* sten_s, const fc_csr* by_source, void* gx, int32_t N,
                       int32_t I, int32_t R, int32_t B, int32_t dtype, void* stream);
/* The contractions of the run-time path (and TangentLin in double precision): a complex GEMM on the matrix pipe,
 *   C[m, n] = alpha * sum_k A[m*sam + k*sak] * op(B[k*sbk + n*sbn]),   op = conj when conj_b != 0,
 * C (M, N) row-major and contiguous, A and B complex arrays addressed with element strides (in complex numbers), all three of
 * dtype FC_F32 (interleaved float pairs) or FC_F64 (double pairs).  With contrib viewed as (n, K = I*R*F) and W_eff as (O, K):
 *   y         = contrib . W_eff^T / F         fc_cgemm(contrib, W, y,  n, O, K,  K, 1,  1, K,  0, 1/F, ...)
 *   g_contrib = gy . conj(W_eff) / F          fc_cgemm(gy, W, gc,      n, K, O,  O, 1,  K, 1,  1, 1/F, ...)
 *   gW_eff    = gy^T . conj(contrib) / F      fc_cgemm(gy, contrib, gW, O, K, n,  1, O,  K, 1,  1, 1/F, ...)            */
/* workspace (optional): fc_cgemm_workspace_bytes(M, N, K, dtype) bytes -- non-zero for products with a small output and a long
 * contraction (the two weight-gradient products: K = the vertex count), which are then split along k over many workgroups with per-slice
 * partials summed in slice order; NULL / 0: one workgroup per 64 x 64 output tile walks all of K. */
size_t fc_cgemm_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t dtype);
int fc_cgemm(const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn,
             int32_t conj_b, double alpha, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ---- TangentLin.forward, reference nn/tangent_lin.py:27-29 ------------------------------- *
 * y[n,o] = sum_i x[n,i] (Re + i Im)[o,i];  re_w, im_w are (O,I) fp32 row-major. */
int fc_tangent_lin_forward(const float* x, const float* re_w, const float* im_w, float* y,
                           int32_t N, int32_t I, int32_t O, void* stream);
/* gx (N,I) c64, g_re / g_im (O,I) fp32 (all overwritten); I, O <= 64.
 * workspace: fc_tangent_lin_backward_workspace_bytes() bytes of per-wavefront partial sums. */
size_t fc_tangent_lin_backward_workspace_bytes(int32_t N, int32_t I, int32_t O);
int fc_tangent_lin_backward(const float* x, const float* gy, const float* re_w, const float* im_w,
                            float* gx, float* g_re, float* g_im, void* workspace, size_t workspace_bytes,
                            int32_t N, int32_t I, int32_t O, void* stream);

/* ---- TangentNonLin.forward (modReLU), reference nn/tangent_nonlin.py:24-35 --------------- *
 * y = relu(|x| + b_c) x/|x| outside the origin box, x inside it.  bias: C floats. */
int fc_tangent_nonlin_forward(const float* x, const float* bias, float* y, int32_t N, int32_t C, void* stream);
/* gx (N,C) c64 overwritten; gbias (C floats) overwritten.
 * workspace: fc_tangent_nonlin_backward_workspace_bytes() bytes of per-block bias partials. */
size_t fc_tangent_nonlin_backward_workspace_bytes(int32_t N, int32_t C);
int fc_tangent_nonlin_backward(const float* x, const float* bias, const float* gy, float* gx, float* gbias,
                               void* workspace, size_t workspace_bytes, int32_t N, int32_t C, void* stream);
/* The same VJP without the second launch: gx as above, the per-group partial sums of the bias gradient left in `workspace` as
 * (fc_tangent_nonlin_backward_groups(N), C) floats for fc_filter_params::bias_partials (a modReLU fused behind a convolution: its
 * backward pass ends in that convolution's finish launch anyway). */
int32_t fc_tangent_nonlin_backward_groups(int32_t N);
int fc_tangent_nonlin_backward_partial(const float* x, const float* bias, const float* gy, float* gx, void* workspace,
                                       size_t workspace_bytes, int32_t N, int32_t C, void* stream);

/* softAbs, reference utils/field.py:29-37 (what ECHOBlock feeds its linear residual, nn/echo_block.py:103): y = |x| outside the origin
 * box, 0 inside; its VJP gx = gy x/|x| (0 inside).  x: count complex64 values, y / gy: count floats. */
int fc_soft_abs_forward(const float* x, float* y, size_t count, void* stream);
int fc_soft_abs_backward(const float* x, const float* gy, float* gx, size_t count, void* stream);

/* The same in double precision (complex128 features, float64 bias): the reference's modules run under .double().
 * TangentLin in double precision is fc_cgemm with Wc = Re + i Im built by the caller. */
int fc_tangent_nonlin_forward_f64(const double* x, const double* bias, double* y, int32_t N, int32_t C, void* stream);
size_t fc_tangent_nonlin_backward_workspace_bytes_f64(int32_t N, int32_t C);
int fc_tangent_nonlin_backward_f64(const double* x, const double* bias, const double* gy, double* gx, double* gbias,
                                   void* workspace, size_t workspace_bytes, int32_t N, int32_t C, void* stream);

/* ---- ECHO descriptors, reference nn/echo.py:94-148 (ECHO.forward with rasterize :30-61, diskMap :11-27) ---- *
 * hist[n,c,b] = sum over in-edges e of n of the bilinear votes of the point ln[e] * exp(-i angle(x[src_e,c])) in the
 * rasterised disk of (2*n_bins+1)^2 cells, each vote carrying x[src_e,c] * wxp[e]; zero features do not vote; the
 * descriptor is |hist| (zero-safe).  x (N,C) c64, any C; ln_t / wxp_t (E) c64 in by_target slot order; hist
 * (N,C,dS) c64 and desc (N,C,dS) f32 are overwritten, dS = fc_echo_hist_dim(n_bins), 1 <= n_bins <= 8.  The channels are
 * independent: a call launches ceil(C / fc_echo_channel_block(n_bins)) channel blocks (64 up to n_bins = 4, then 57, 42, 30, 23: one
 * channel per lane, the histograms of a workgroup in LDS).  Any n_bins, and double precision: fc_echo_forward_generic below.
 * Backward: by_source groups the edges by source (nbr = targets), ln_s / wxp_s in that slot order; g_desc (N,C,dS) f32;
 * gx (N,C) c64 is overwritten; hist_grad_workspace: N*C*dS complex64 values of scratch. */
int fc_echo_hist_dim(int32_t n_bins);
int fc_echo_channel_block(int32_t n_bins);
int fc_echo_forward(const float* x, const float* ln_t, const float* wxp_t, const fc_csr* by_target, float* hist,
                    float* desc, int32_t N, int32_t E, int32_t C, int32_t n_bins, void* stream);
int fc_echo_backward(const float* x, const float* ln_s, const float* wxp_s, const fc_csr* by_source, const float* hist,
                     const float* g_desc, float* gx, float* hist_grad_workspace, int32_t N, int32_t E, int32_t C,
                     int32_t n_bins, void* stream);

/* ---- TransField (the learned 'gradient' of LiftBlock), reference nn/trans_field.py:78-113, weightContrib* :9-24 ---- *
 * x (N,Cin) f32 scalar features, Cin <= 4; lift_sten: the stencil columns m = 0, 1 (reference segmentation.ipynb:204)
 * in ORIGINAL edge order, element (e,r,j) at complex index (e*R + r)*sten_stride + j -- sten_stride = 2 for a packed
 * (E,R,2) array, 2B+1 for a pointer to column m = 0 of the full (E,R,2B+1) stencil, 0 when lift_sten is the (E,8) factor table of
 * fc_precomp_graph (the two columns are then formed on the fly: w_r c and w_r c e^{i theta}; no (E,R,2) array exists); by_target / by_source: the edges grouped by target / source with
 * slot_to_edge (E) int64 giving the original edge of every slot; zonal_ang, zonal_mag (O,Cin,R) f32, phase (O,Cin) f32
 * (zeros for ftype 0); y (N,O) c64, O <= 64.  The forward call also leaves ang (N,Cin,R) c64, mag (N,Cin,R) f32 and
 * s1sum (N,R) c64 for the backward call, which overwrites gx (N,Cin) f32 and the parameter gradients (g_phase for
 * ftype != 0 only) using a caller-provided workspace. */
int fc_trans_field_forward(const float* x, const float* lift_sten, const fc_csr* by_target, const int64_t* slot_to_edge,
                           const float* zonal_ang, const float* zonal_mag, const float* phase, float* y, float* ang,
                           float* mag, float* s1sum, int32_t N, int32_t E, int32_t Cin, int32_t O, int32_t R, int32_t sten_stride,
                           void* stream);
size_t fc_trans_field_backward_workspace_bytes(int32_t N, int32_t Cin, int32_t O, int32_t R);
int fc_trans_field_backward(const float* lift_sten, const fc_csr* by_source, const int64_t* slot_to_edge_s,
                            const float* zonal_ang, const float* zonal_mag, const float* phase, const float* ang,
                            const float* mag, const float* s1sum, const float* gy, float* gx, float* g_zonal_ang,
                            float* g_zonal_mag, float* g_phase, void* workspace, size_t workspace_bytes, int32_t N,
                            int32_t E, int32_t Cin, int32_t O, int32_t R, int32_t sten_stride, int32_t ftype, void* stream);

/* ---- whole blocks from ONE call per pass ---------------------------------------------------------------- *
 * The reference's networks are built from three blocks -- FCResNetBlock (nn/fc_resnet_block.py:65-88), ECHOBlock
 * (nn/echo_block.py:73-103) and LiftBlock (nn/lift_block.py:35-55) -- and train with batch size 1 on a different ~1k-vertex mesh every
 * step (segmentation.ipynb:120,137): a binding that crosses into the library once per KERNEL spends more host time per block than the
 * GPU needs to run it.  The entry points below enqueue a whole block pass -- filter assembly, convolutions with their fused
 * residual / modReLU epilogues, the TangentLin residual, the descriptor splat, and in the backward pass the complete VJP chain with
 * every parameter gradient -- from one foreign call, with caller-owned buffers only:
 *   saved      what the backward pass needs from the forward pass (pre-activations, activations, backward filter images, ...):
 *              fc_*_saved_bytes() bytes, 256-byte aligned, written by *_forward, read by *_backward, untouched in between
 *   workspace  scratch of one call: fc_*_workspace_bytes(..., backward) bytes, 256-byte aligned
 * Same kernels and the same results, bit for bit, as the per-operator calls above in the order the reference applies them. */
typedef struct fc_mesh {           /* one mesh's support graph as the convolutions consume it */
    int32_t N, E, R, B;            /* vertices, support edges, n_rings, band_limit of the stencil */
    int32_t kind;                  /* 0 dense stencil rows, 1 factored records, 2 geometric forward records (factored backward) */
    const fc_csr* by_target;       /* as for fc_forward* (nbr also read by the ECHO / lift blocks) */
    const fc_csr* by_source;       /* as for fc_backward_data* */
    const float* fwd;              /* sten_t / rec_t / geo_t, by kind, in by_target slot order */
    const float* bwd;              /* sten_s (kind 0) or rec_s, in by_source slot order */
    int32_t mode;                  /* fc_mfma_mode of the block's convolutions (0: the default) */
} fc_mesh;

/* FCResNetBlock: out = modReLU_2(res(x) + conv2(modReLU_1(conv1(x)))), reference nn/fc_resnet_block.py:84-88.
 * conv1: (C_mid, C_in) filter, conv2: (C_out, C_mid); bias1 (C_mid), bias2 (C_out): the TangentNonLin biases; res_re / res_im
 * (C_out, C_in): the TangentLin residual.  The g_* members receive the gradients in the backward call (conv*.g_*, g_bias*, g_res_*);
 * the bias riders of conv1 / conv2 (fc_filter_params::bias_partials ...) are set by the library and ignored on entry. */
typedef struct fc_resnet_block_params {
    int32_t C_in, C_mid, C_out;
    fc_filter_params conv1, conv2;
    const float* bias1;
    const float* bias2;
    const float* res_re;
    const float* res_im;
    float* g_bias1;
    float* g_bias2;
    float* g_res_re;
    float* g_res_im;
} fc_resnet_block_params;
size_t fc_resnet_block_saved_bytes(const fc_mesh* mesh, const fc_resnet_block_params* p);
size_t fc_resnet_block_workspace_bytes(const fc_mesh* mesh, const fc_resnet_block_params* p, int32_t backward);
/* x (N,C_in) c64 -> out (N,C_out) c64 */
int fc_resnet_block_forward(const float* x, const fc_mesh* mesh, const fc_resnet_block_params* p, float* out, void* saved,
                            size_t saved_bytes, void* workspace, size_t workspace_bytes, void* stream);
/* g_out (N,C_out) c64 -> gx (N,C_in) c64 and every parameter gradient of the block */
int fc_resnet_block_backward(const float* x, const float* g_out, const fc_mesh* mesh, const fc_resnet_block_params* p, const void* saved,
                             size_t saved_bytes, float* gx, void* workspace, size_t workspace_bytes, void* stream);

/* The tangent-feature half of ECHOBlock: desc = ECHO(modReLU(conv(x))), reference nn/echo_block.py:93-94 (the MLP on the descriptors
 * and the linear residual on |x|, :95-103, are the reference's own nn.Linear layers and stay with the caller).  conv: (n_des, C_in)
 * filter; bias: the first n_des entries of the module's TangentNonLin bias (:57,93); ln_* / wxp_* (E) c64 in by_target / by_source slot
 * order as for fc_echo_forward / fc_echo_backward; desc (N, n_des, fc_echo_hist_dim(n_bins)) f32. */
typedef struct fc_echo_block_params {
    int32_t C_in, n_des, n_bins;
    fc_filter_params conv;
    const float* bias;
    float* g_bias;
} fc_echo_block_params;
size_t fc_echo_block_saved_bytes(const fc_mesh* mesh, const fc_echo_block_params* p);
size_t fc_echo_block_workspace_bytes(const fc_mesh* mesh, const fc_echo_block_params* p, int32_t backward);
int fc_echo_block_forward(const float* x, const fc_mesh* mesh, const float* ln_t, const float* wxp_t, const fc_echo_block_params* p,
                          float* desc, void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, void* stream);
int fc_echo_block_backward(const float* x, const float* g_desc, const fc_mesh* mesh, const float* ln_s, const float* wxp_s,
                           const fc_echo_block_params* p, const void* saved, size_t saved_bytes, float* gx, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ECHOBlock behind its descriptors, reference nn/echo_block.py:95-103:  y = lin3(relu(lin2(relu(lin1(d))))) + res(softAbs(x)).
 * d (N, D) f32 the flattened descriptors (D = n_des * fc_echo_hist_dim(n_bins)), x (N, C_in) c64 the block's input, y (N, C_out) f32;
 * w1 (H1, D), w2 (H2, H1), w3 (C_out, H2), wr (C_out, C_in) and the biases as torch.nn.Linear holds them (the reference: H1 = 128,
 * H2 = 64); H1 <= 128, H2 <= 64, C_in <= 64, C_out <= 64, else FC_ERR_UNSUPPORTED (the host then composes the tail of dense layers).
 * forward: h1 (N, H1) and h2 (N, H2) receive the hidden activations (after their ReLUs) -- the caller keeps them, with d and x, for the
 * backward pass.  backward: g (N, C_out) the cotangent of y; g_d (N, D), gx (N, C_in) c64, every g_* of the struct (weights and biases)
 * and g_h1 (N, H1, scratch the caller provides) are written.  All products run on v_mfma_f32_16x16x4_f32 (fp32 operands and sums);
 * three launches per pass (csrc/fc_head.hip). */
typedef struct fc_echo_head_params {
    int32_t D, H1, H2, C_in, C_out;
    const float *w1, *b1, *w2, *b2, *w3, *b3, *wr, *br;
    float *g_w1, *g_b1, *g_w2, *g_b2, *g_w3, *g_b3, *g_wr, *g_br;       /* backward only */
} fc_echo_head_params;
size_t fc_echo_head_forward_workspace_bytes(int32_t N, const fc_echo_head_params* p);
int fc_echo_head_forward(const float* d, const float* x, const fc_echo_head_params* p, float* h1, float* h2, float* y, void* workspace,
                         size_t workspace_bytes, int32_t N, void* stream);
size_t fc_echo_head_backward_workspace_bytes(int32_t N, const fc_echo_head_params* p);
int fc_echo_head_backward(const float* d, const float* x, const float* h1, const float* h2, const float* g, const fc_echo_head_params* p,
                          float* g_d, float* gx, float* g_h1, void* workspace, size_t workspace_bytes, int32_t N, void* stream);

/* LiftBlock: out = modReLU(TransField(x)), reference nn/lift_block.py:53-55.  x (N,C_in) f32, C_in <= 4, C_out <= 64; lift_sten /
 * sten_stride / slot_to_edge_* as for fc_trans_field_forward / _backward (mesh: N, E, R and the two groupings with nbr; kind, fwd, bwd
 * are not read); phase / g_phase for ftype != 0 only (phase: zeros otherwise). */
typedef struct fc_lift_block_params {
    int32_t C_in, C_out, ftype;
    const float* zonal_ang;
    const float* zonal_mag;
    const float* phase;
    const float* bias;
    float* g_zonal_ang;
    float* g_zonal_mag;
    float* g_phase;
    float* g_bias;
} fc_lift_block_params;
size_t fc_lift_block_saved_bytes(const fc_mesh* mesh, const fc_lift_block_params* p);
size_t fc_lift_block_workspace_bytes(const fc_mesh* mesh, const fc_lift_block_params* p, int32_t backward);
int fc_lift_block_forward(const float* x, const float* lift_sten, int32_t sten_stride, const fc_mesh* mesh, const int64_t* slot_to_edge_t,
                          const fc_lift_block_params* p, float* out, void* saved, size_t saved_bytes, void* stream);
int fc_lift_block_backward(const float* g_out, const float* lift_sten, int32_t sten_stride, const fc_mesh* mesh, const int64_t* slot_to_edge_s,
                           const fc_lift_block_params* p, const void* saved, size_t saved_bytes, float* gx, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- TransField and ECHO outside the specialised kernels' range: any number of scalar inputs / output channels / rings / raster bins,
 * float32 or float64 (dtype: fc_dtype of every tensor; the reference's TransField / LiftBlock run under .double(), nn/trans_field.py:78-113;
 * its ECHO and FCPrecomp do not -- both raise a dtype error -- so the float64 ECHO here has no reference counterpart).  Run-time loops, one
 * thread per output entry, fixed summation order: a correctness path.  Same arguments as the specialised calls; lift_sten is a complex array
 * (float2 / double2 elements) with sten_stride >= 2 (no factor table); the backward workspace is
 * fc_trans_field_backward_generic_workspace_bytes(...) bytes; the ECHO calls take fc_echo_generic_workspace_bytes(n_bins) bytes of scratch
 * (the raster map) and dS = fc_echo_hist_dim_generic(n_bins). */
int fc_trans_field_forward_generic(const void* x, const void* lift_sten, const fc_csr* by_target, const int64_t* slot_to_edge,
                                   const void* zonal_ang, const void* zonal_mag, const void* phase, void* y, void* ang, void* mag,
                                   void* s1sum, int32_t N, int32_t E, int32_t Cin, int32_t O, int32_t R, int32_t sten_stride, int32_t dtype,
                                   void* stream);
size_t fc_trans_field_backward_generic_workspace_bytes(int32_t N, int32_t Cin, int32_t O, int32_t R, int32_t dtype);
int fc_trans_field_backward_generic(const void* lift_sten, const fc_csr* by_source, const int64_t* slot_to_edge_s, const void* zonal_ang,
                                    const void* zonal_mag, const void* phase, const void* ang, const void* mag, const void* s1sum,
                                    const void* gy, void* gx, void* g_zonal_ang, void* g_zonal_mag, void* g_phase, void* workspace,
                                    size_t workspace_bytes, int32_t N, int32_t E, int32_t Cin, int32_t O, int32_t R, int32_t sten_stride,
                                    int32_t ftype, int32_t dtype, void* stream);
int fc_echo_hist_dim_generic(int32_t n_bins);
size_t fc_echo_generic_workspace_bytes(int32_t n_bins);
int fc_echo_forward_generic(const void* x, const void* ln_t, const void* wxp_t, const fc_csr* by_target, void* hist, void* desc,
                            void* workspace, size_t workspace_bytes, int32_t N, int32_t E, int32_t C, int32_t n_bins, int32_t dtype,
                            void* stream);
int fc_echo_backward_generic(const void* x, const void* ln_s, const void* wxp_s, const fc_csr* by_source, const void* hist,
                             const void* g_desc, void* gx, void* hist_grad_workspace, void* workspace, size_t workspace_bytes, int32_t N,
                             int32_t E, int32_t C, int32_t n_bins, int32_t dtype, void* stream);

/* ---- support-graph build (what every FieldConv needs before its first launch on a mesh) ---------------------------- *
 * From the operator's own inputs (reference nn/field_conv.py:104-121): supp_edges (E,2) int64, col 0 = source, col 1 =
 * target; supp_sten (E,R,F) c64 contiguous, F = 2B+1 (NULL: edge grouping only, for the ECHO / TransField entry points)
 * to the two fc_csr groupings and the per-edge records of fc_forward_factored / fc_forward_geometric /
 * fc_backward_data_factored.  Slots are ordered by (vertex, lower ring q, original edge index); perm_* (E) int64 give the
 * original edge of every slot; runs_* are (N,8) int32; rec_* hold E rows of fc_factored_record_floats(B) floats, geo_t E
 * rows of 8 floats -- the caller allocates and zeroes the >= 1 KiB of padding behind them.  flags (one device int32):
 * bit 0 the stencil is NOT of the rank-1 / two-adjacent-rings form (every row is reconstructed and compared, relative
 * tolerance 2e-6): rec_* / geo_t must not be used, use the dense entry points with supp_sten[perm]; bit 1 the phases are
 * not geometric in the frequency: geo_t must not be used; bit 2 an edge refers to a vertex outside [0, N).
 * No host synchronisation inside; reading `flags` is the caller's one synchronisation per mesh. */
size_t fc_graph_workspace_bytes(int32_t N, int32_t E, int32_t R, int32_t F, int32_t with_stencil);
int fc_graph_build(const int64_t* supp_edges, const float* supp_sten, int32_t N, int32_t E, int32_t R, int32_t F,
                   int32_t* rowptr_t, int32_t* nbr_t, int32_t* runs_t, int64_t* perm_t, int32_t* rowptr_s,
                   int32_t* nbr_s, int32_t* runs_s, int64_t* perm_s, float* rec_t, float* rec_s, float* geo_t,
                   int32_t* flags, void* workspace, size_t workspace_bytes, void* stream);

/* ---- FCPrecomp (stencil assembly), reference transforms/fc_precomp.py:53-97 ------------------------------------------ *
 * Inputs as the reference's data object holds them: log_mag (E) f32, log_ang (E) f32, xp (E) c64, w (N) f32 (vertex
 * areas), supp_edges (E,2) int64, epsilon (support radius).  Two steps because the number of kept edges E' (those with
 * log_mag / epsilon <= 1) sizes the outputs: fc_precomp_mark enqueues the selection; the two int32 at
 * fc_precomp_kept_count_ptr() then hold E' and a flag that is non-zero when a kept edge refers to a vertex outside [0, N)
 * (device memory: the caller's one synchronisation; supp_edges may be NULL for fc_precomp_mark: no range check).  Then, with
 * the same workspace, EITHER
 *   fc_precomp_build   writes supp_edges_out (E',2) int64, supp_sten (E',R,F) c64, ln (E') c64 and wxp (E') c64 in the
 *                      original edge order (the reference's outputs as they are), OR
 *   fc_precomp_graph   goes straight to what the convolutions consume: supp_edges_out, ln, wxp as above,
 *                      `factors` (E',8) f32 = [q bits, w_q, w_{q+1}, 0, Re wxp, Im wxp, cos theta, sin theta] per kept edge
 *                      (the stencil row is w_r * wxp * e^{i m theta}: anything that wants dense rows builds them from this),
 *                      and every output of fc_graph_build (both groupings, ring-run offsets, permutations, factored and
 *                      geometric records) -- the (E',R,F) stencil is never written or read.  graph_workspace:
 *                      fc_graph_workspace_bytes(N, E', R, F, 1) bytes; flags as for fc_graph_build (bits 0/1 never set).
 *                      rec_t / rec_s hold E' + rec_pad_rows rows, geo_t E' + geo_pad_rows: the rows behind the E'-th are
 *                      zero-filled here (the convolution kernels stream a little past the end of the records). */
size_t fc_precomp_workspace_bytes(int32_t N, int32_t E);
int fc_precomp_mark(const float* log_mag, const int64_t* supp_edges, float epsilon, int32_t N, int32_t E, void* workspace,
                    size_t workspace_bytes, void* stream);
const int32_t* fc_precomp_kept_count_ptr(const void* workspace, int32_t E);
int fc_precomp_build(const float* log_mag, const float* log_ang, const float* xp, const float* w, const int64_t* supp_edges,
                     float epsilon, int32_t N, int32_t E, int32_t R, int32_t F, int64_t* supp_edges_out, float* supp_sten,
                     float* ln, float* wxp, void* workspace, size_t workspace_bytes, void* stream);
int fc_precomp_graph(const float* log_mag, const float* log_ang, const float* xp, const float* w, const int64_t* supp_edges,
                     float epsilon, int32_t N, int32_t E, int32_t E_kept, int32_t R, int32_t F, int32_t rec_pad_rows,
                     int32_t geo_pad_rows, int64_t* supp_edges_out, float* ln, float* wxp, float* factors, int32_t* rowptr_t, int32_t* nbr_t, int32_t* runs_t, int64_t* perm_t,
                     int32_t* rowptr_s, int32_t* nbr_s, int32_t* runs_s, int64_t* perm_s, float* rec_t, float* rec_s, float* geo_t,
                     int32_t* flags, void* precomp_workspace, size_t precomp_workspace_bytes, void* graph_workspace,
                     size_t graph_workspace_bytes, void* stream);

/* ---- fused Adam over a flat float32 parameter buffer (torch.optim.Adam arithmetic, L2 weight decay, no amsgrad) -------- *
 * params, grads, exp_avg, exp_avg_sq: n floats each, n a multiple of 4, 16-byte aligned; step: one device float holding
 * the number of steps taken so far (incremented on the device, so the call is capturable in a HIP graph). */
int fc_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* step, size_t n, float lr,
                 float beta1, float beta2, float eps, float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FIELDCONV_HIP_H */
