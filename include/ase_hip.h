/* ase_hip.h — C ABI of libase_hip.so: the MI355X (gfx950) kernels behind the ASE/AMP PPO update.
 *
 * The reference (nv-tlabs/ASE) is pure Python on stock PyTorch ops; it has no FFI.  Each entry
 * point below therefore cites the reference *Python* statement(s) it replaces (paths relative to
 * /root/reference/ase/).  INTEGRATION.md shows the ctypes binding a maintainer of the reference
 * would add for each.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ASE_E* code otherwise
 *     (ase_hip_last_error() gives the text); nothing allocates, frees or synchronises;
 *     work is enqueued on `stream` (a hipStream_t), so calls are hipGraph-capturable.
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS.
 *   - `dtype` selects the storage type of activations / shadow weights that feed the matrix
 *     cores: ASE_F32 (exact f32 MFMA, parity mode), ASE_BF16 (bf16 MFMA, f32 accumulate) or — for the two GEMM
 *     entry points only — ASE_F32X3 (f32 storage like ASE_F32, each product as three bf16 MFMAs on a hi/lo split) and, for
 *     ase_hip_gemm_nt, ASE_F32H3 (three f16 MFMAs on a hi/lo split of power-of-two scaled operands: ~22 significant bits).
 *     Running statistics are f64, master weights / gradients / Adam state / loss math are f32.
 *   - GEMM operands live in "padded" buffers: K (the contracted, contiguous dimension) is a
 *     multiple of 128 bytes / sizeof(type) and the padding is zero.
 *   - row index maps: a logical minibatch row r reads dataset row p = idx ? idx[r] : r; when
 *     remap_h > 0 the dataset is the time-major experience buffer [H=remap_h, N=remap_n, ...] and
 *     p is the env-major flat index (p = env*H + t, rl_games swap_and_flatten01), i.e. the
 *     physical row is (p % H) * N + p / H.  This is how learning/ase_agent.py:108-113 and
 *     learning/amp_datasets.py:14-27 are applied without materialising copies.
 */
#ifndef ASE_HIP_H
#define ASE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASE_HIP_ABI_VERSION 7

enum { ASE_F32 = 0, ASE_BF16 = 1, ASE_F32X3 = 2 /* f32 storage, products as 3 bf16 MFMAs on a hi/lo split (GEMMs only) */,
       ASE_F32H3 = 4 /* 4-byte storage, products as 3 f16 MFMAs on hi/lo splits of operands scaled by 2^ea / 2^eb (the exponents ride in
                        bits 8-15 / 16-23 of dtype: ASE_F32H3 | (ea << 8) | (eb << 16)).  ase_hip_gemm_nt: A is plain f32 (split in the
                        kernel), B is the PACKED SPLIT format ase_hip_refresh_shadow(dtype = ASE_F32H3 | eb << 16) writes - per group of
                        8 consecutive k: 8 hi halves, then 8 lo halves, of W * 2^eb (32 bytes, leading dimension as for f32).  Operands
                        must satisfy |x| * 2^e < 65504 (overflow turns the output into NaN); best accuracy for |x| * 2^e >= 2^-2 */,
       ASE_F16 = 3 /* IEEE half storage + v_mfma_f32_32x32x16_f16, f32 accumulate: what the reference's mixed_precision flag
                      (torch.cuda.amp autocast + GradScaler, learning/ase_agent.py:216,271-288) computes in; conversions saturate */ };
/* activations: the names of rl_games' activations_factory (learning/ase_network_builder.py:162); swish = SiLU */
enum { ASE_ACT_NONE = 0, ASE_ACT_RELU = 1, ASE_ACT_TANH = 2, ASE_ACT_SILU = 3, ASE_ACT_ELU = 4, ASE_ACT_GELU = 5,
       ASE_ACT_SIGMOID = 6, ASE_ACT_SELU = 7, ASE_ACT_SOFTPLUS = 8 };
enum { ASE_AUX_NONE = 0, ASE_AUX_RELU_MASK = 1, ASE_AUX_TANH_GRAD = 2, ASE_AUX_RELU_BITS = 3 /* aux = bit matrix written by mask_out */,
       ASE_AUX_PREACT = 4 /* aux = the layer's PRE-activation z (dtype, written by the forward launch as its twin): multiply by
                             act'(z); the activation id rides in bits 8+ of aux_mode: ASE_AUX_PREACT | (ASE_ACT_x << 8) */ };
enum { ASE_OK = 0, ASE_EINVAL = -1, ASE_ELAUNCH = -2, ASE_EUNSUPPORTED = -3 };

int ase_hip_abi_version(void);
const char* ase_hip_last_error(void);

/* Which kernel ase_hip_gemm_nt launches for a shape (host only): 0 = 64 x 64 tile (narrow outputs), 1 = 128 x 128,
 * 2 = phased 256 x 256 (16-bit storage), 3 = lock-step 256 x 256 (4-byte storage), 4 = 64 x 128, 5 = 64 x 64 with
 * 128-byte rows, 6 = phased 192 x 256 (16-bit storage, single rounds of 192-255 tiles; falls back to 2 when the
 * launch needs the LDS-slab epilogue).  bench.py uses it to attribute launch times to the dominant kernel. */
int ase_hip_gemm_nt_kernel_id(int M, int N, int K, int dtype);

/* Kernel-tuning aid (scripts/lab): when buf is a device uint64[4 * workgroups] array, the phased NT kernel stamps
 * {entry, first tile landed, main loop done, stores retired} per workgroup (100 MHz clock); NULL switches it off. */
int ase_hip_debug_nt_profile(void* buf);
/* ... shader_clock != 0: the two main-loop stamps (1, 2) of the phased kernel count SHADER clocks (s_memtime) instead of the
 * 100 MHz clock - two profiled launches of one shape give the clock frequency the chip actually sustains under that kernel
 * (bench.py `roofline.sustained_clock_mhz`: MI355X does not hold its 2.4 GHz boost under dense MFMA load). */
int ase_hip_debug_nt_profile_clock(int shader_clock);

/* ---------------------------------------------------------------------------------------------
 * Dense layers (matrix cores).
 * ------------------------------------------------------------------------------------------- */

/* C[m,n] = mask( act( alpha * sum_k A[m,k] * B[n,k] + bias[n] ) )       "NT" GEMM
 *   A [M,K] dtype, B [N,K] dtype (weights, K contiguous), C [M,N] dtype or f32 (out_f32).
 *   aux [M,N] dtype: ASE_AUX_RELU_MASK multiplies by (aux > 0), ASE_AUX_TANH_GRAD by (1 - aux^2);
 *   rows m >= aux_split (> 0) read aux row m - aux_delta (a stacked block of rows re-using another block's mask:
 *   the gradient-penalty chain rides on the discriminator's data-gradient launches).
 *   colsum (nullable, f32[colsum_n]) += column sums of the stored values for n < colsum_n (atomic):
 *   the bias gradient of the producing layer.
 *   mask_out (nullable) = the layer's TWIN output, what the data-gradient launch of the same activation will read as aux:
 *   act NONE / RELU / TANH: uint32 [M, ldmask], N % 32 == 0: bit n % 32 of word [m, n / 32] = (stored C[m,n] > 0) - read back
 *     with ASE_AUX_RELU_BITS (ldaux in words): 1/16 of the bytes of re-reading the 16-bit activation in the store-bound epilogue;
 *   act >= ASE_ACT_SILU: dtype [M, ldmask] (ldmask in ELEMENTS), the pre-activation z = alpha * A.B^T + bias - read back with
 *     ASE_AUX_PREACT (the derivative of a non-monotonic activation is not a function of its output; tanh keeps
 *     ASE_AUX_TANH_GRAD on the output itself).
 *   alpha_dev (nullable; ABI 6, a RECORD since ABI 7): DEVICE f32[2] scale record {factor, overflow count}.  The launch multiplies
 *   alpha by `factor` when it RUNS - how the factors of the dynamic loss scale (ase_hip_scaler_step's table: S, 1 / S, 1 / S^2, 1) reach
 *   launches recorded once and replayed across scale changes - and ADDS to `count` (by an unspecified positive amount) when an element
 *   it STORED into C / mask_out's pre-activation twin is not finite or sits at the storage type's saturation value (ASE_F16 conversions
 *   saturate at +-65504 where autocast would produce inf): GradScaler's found_inf, detected by the producer instead of a pass that
 *   re-reads every buffer of the step (ase_hip_scaler_check).  The same convention for every `*_dev` argument below; the loss heads
 *   report their stored head gradients, ase_hip_gemm_tn(_grouped) and ase_hip_sqnorm only read the factor.
 * Replaces: nn.Linear + activation forward  (learning/ase_network_builder.py:255-259,305-324,
 *   learning/amp_network_builder.py:81-84), and autograd's data-gradient of the same layers
 *   (B = the transposed weight shadow; mask = derivative of the previous activation), and the
 *   transposed-MLP chain of the gradient penalty (learning/amp_agent.py:453-459). */
int ase_hip_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                    const float* bias, const void* aux, int64_t ldaux, int aux_split, int aux_delta,
                    float* colsum, int colsum_n, void* mask_out, int64_t ldmask, int M, int N, int K, int act,
                    int aux_mode, int out_f32, float alpha, float* alpha_dev, int dtype, void* stream);

/* G[n, kmap(k)] += alpha * sum_m A[m,n] * B[m,k]   for n < n_real, kmap(k) valid     "TN" GEMM
 *   A [M,N] dtype (output gradients), B [M,K] dtype (layer inputs), G f32 [n_real, k_real]
 *   (master-layout weight gradient, accumulated with f32 atomics; split over M internally).
 *   kmap undoes the padded-concat layout of the first actor/critic layer: k < split_src -> k;
 *   k >= split_dst -> k - (split_dst - split_src); columns in between are padding.
 *   (layers without a concat pass split_src = split_dst = k_real).
 *   gbias (nullable, f32[n_real]) += alpha * sum_{m < bias_rows} A[m,n] (bias_rows <= 0: all rows): the bias gradient of the same layer, reduced from
 *   the A tiles the kernel stages anyway (a handful of atomics per column instead of one per row tile).
 * Replaces: autograd's weight gradient of nn.Linear inside loss.backward()
 *   (learning/ase_agent.py:271). */
int ase_hip_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* G, float* gbias,
                    int bias_rows, int M, int N, int K, int n_real, int k_real, int split_src, int split_dst,
                    float alpha, const float* alpha_dev, int dtype, void* stream);

/* All weight gradients of one branch of an optimisation step in ONE grouped launch (16-bit storage).  They only depend on
 * buffers the data-gradient chain has already written, and one grid of ~256 long contractions pays the split-M reduction once
 * per step instead of once per layer.
 *   problems: int64[n][16] = {A, lda, B, ldb, G, gbias (or 0), bias_rows (0 = all), M, N, K, n_real, k_real, split_src,
 *             split_dst, alpha (f32 bit pattern), 0}, fields as in ase_hip_gemm_tn, leading dimensions in elements;
 *             M and bias_rows multiples of 64.
 *   ase_hip_gemm_tn_grouped_plan (host only, no GPU needed): validates the HOST copy of the table, fills field 15 (tiles
 *             along k; bit 30: the gradient buffer is shared with another problem of the launch) and writes the work list
 *             int32[n_work][4] = {problem, 256 x 256 output tile, first row, 64-row K-tiles | workspace slab << 16} in LAUNCH
 *             order - positions [x n_work/8, (x+1) n_work/8) run on XCD x: the tiles of one (problem, row range) share their
 *             operand panels through that XCD's L2, so such groups are bin-packed whole into the 8 ranges and the ranges
 *             padded with empty items (K-tiles = 0; n_work is a multiple of 8) - and (red non-null) the reduce list
 *             int32[n_red][4] = {problem, tile, first slab, splits} (split s of a tile: slab first + s * tiles of the problem);
 *             target_wg <= 0: one workgroup per CU.
 *   ase_hip_gemm_tn_grouped: launch with DEVICE copies of the planned tables.  workspace (device f32, 16-byte aligned,
 *             n_work * ASE_TN_SLAB floats, private to the launch until it completes): every work item stores its partial
 *             tile there with plain 16-byte stores and a second kernel adds the sums into G / gbias (deterministic, no
 *             atomics: memory-side f32 atomics run at ~1.4 TB/s on this chip, 64 MB of them per launch).  workspace NULL:
 *             the work items add into G with f32 atomics directly.
 * Replaces: loss.backward()'s weight / bias gradients of every nn.Linear (learning/ase_agent.py:271). */
#define ASE_TN_SLAB (65536 + 256)
int ase_hip_gemm_tn_grouped_plan(int64_t* problems, int n_problems, int target_wg, int32_t* work, int max_work,
                                 int* n_work, int32_t* red, int max_red, int* n_red);
int ase_hip_gemm_tn_grouped(const int64_t* problems, const int32_t* work, int n_work, const int32_t* red, int n_red,
                            float* workspace, const float* alpha_dev, int dtype, void* stream);

/* Shadow copies of one weight matrix for the matrix cores: W_s [n_pad,k_pad] and its transpose
 * Wt_s [k_pad,n_pad] (both dtype, zero padded, concat columns moved to split_dst).  Run after
 * every optimizer step.  (No reference counterpart: the reference multiplies f32 masters.)
 * dtype = ASE_F32H3 | (eb << 16): the packed half-split shadows of W * 2^eb (see ASE_F32H3; buffers sized and strided as f32). */
int ase_hip_refresh_shadow(const float* W, int n_real, int k_real, void* Ws, int64_t ldws,
                           void* Wts, int64_t ldwts, int split_src, int split_dst, int dtype,
                           void* stream);

/* The same for every layer in ONE launch.  desc: DEVICE int64[n_layers][12] =
 * {W, n_real, k_real, Ws, ldws, Wts, ldwts, split_src, split_dst - split_src, bias, bias_shadow, ceil(k_real/32)}
 * (pointers as integers; bias_shadow f32[n_pad] feeds ase_hip_gemm_nt's bias). */
int ase_hip_refresh_shadow_multi(const int64_t* desc, int n_layers, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Running mean/std normaliser (rl_games RunningMeanStd; learning/common_agent.py:49,323-325,
 * learning/amp_agent.py:26,535-538, learning/ase_agent.py:170-181) fused with the minibatch
 * gather (learning/amp_datasets.py:14-27).
 * state: f64[2*D+1] = running_mean[D], running_var[D], count.
 * ------------------------------------------------------------------------------------------- */

/* sums[2*D] (f64, atomic) += { sum_r (x[r,j]-shift[j]), sum_r (x[r,j]-shift[j])^2 }, shift = f32(running_mean) */
int ase_hip_rms_moments(const float* src, int64_t ld_src, int D, const int32_t* idx, int remap_h,
                        int remap_n, int M, const double* state, double* sums, void* stream);

/* The same for up to 4 streams of equal width D and row count M in ONE launch (agent / replay / demo AMP observations,
 * learning/amp_agent.py:280-289): HOST arrays of per-stream sources, index maps and partial-sum buffers. */
int ase_hip_rms_moments_multi(const float* const* srcs, const int64_t* ld_srcs, const int32_t* const* idxs,
                              const int* remap_h, const int* remap_n, double* const* sums, int n_streams, int D, int M,
                              const double* state, void* stream);
int ase_hip_rms_normalize_multi(const float* const* srcs, const int64_t* ld_srcs, const int32_t* const* idxs,
                                const int* remap_h, const int* remap_n, const float* const* means,
                                const float* const* stds, void* const* outs, const int64_t* ld_outs, int n_streams, int D,
                                int M, int dtype, void* stream);

/* Sequentially merge n_streams batches (sums[s][2*D], counts[s] rows each; counts are the GLOBAL
 * row counts) into `state` exactly as RunningMeanStd.forward does in training mode, and after
 * each merge emit mean_f32[s][D] and std_f32[s][D] = sqrt(f32(var)+1e-5).  n_streams == 0 emits
 * one pair from the current state (eval mode). */
int ase_hip_rms_finalize(double* state, int D, const double* sums, const int32_t* counts /* HOST */,
                         int n_streams, float* mean_out, float* std_out, void* stream);

/* out_i[r, col_off_i + j] = clamp((x[map(r), j] - mean[j]) / std[j], -5, 5), i < 3 destinations. */
int ase_hip_rms_normalize(const float* src, int64_t ld_src, int D, const int32_t* idx, int remap_h,
                          int remap_n, int M, const float* mean, const float* std,
                          void* out0, int64_t ld0, void* out1, int64_t ld1, void* out2, int64_t ld2,
                          int dtype, void* stream);

/* y = sqrt(f32(var)+1e-5) * clamp(x,-5,5) + f32(mean): RunningMeanStd(..., unnorm=True)
 * (learning/ase_agent.py:140-141,391-392), D == 1. */
int ase_hip_rms_unnormalize(const double* state, const float* x, float* y, int64_t n, void* stream);

/* Generic row gather + cast: dst[r, 0:D] = src[map(r), 0:D]; dst dtype = dst_dtype
 * (learning/amp_datasets.py:21-22 for the small per-row tensors). */
int ase_hip_gather_rows(const float* src, int64_t ld_src, int D, const int32_t* idx, int remap_h,
                        int remap_n, int M, void* dst, int64_t ld_dst, int dst_dtype, void* stream);

/* Several fields by ONE launch.  desc: DEVICE int64[n_fields][6] = {src, ld_src, D, dst, ld_dst, dst_dtype}.  idx may be NULL
 * (identity row map): the launch is then a multi-tensor f32 -> storage-type conversion (the exact gradient-penalty chain handed
 * to the 16-bit launches, UpdateEngine._gp_f32). */
int ase_hip_gather_multi(const int64_t* desc, int n_fields, const int32_t* idx, int remap_h, int remap_n,
                         int M, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loss heads: forward value + analytic gradient w.r.t. the head inputs, in one pass.
 * `acc` is a device array of f64 partial sums (layout ASE_ACC_* below), zeroed by begin_step.
 * ------------------------------------------------------------------------------------------- */
enum {
    ASE_ACC_MASK_SUM = 0,   /* sum rand_action_mask                         */
    ASE_ACC_A_LOSS,         /* sum mask * ppo surrogate                      */
    ASE_ACC_B_LOSS,         /* sum mask * bound loss                         */
    ASE_ACC_ENTROPY,        /* sum mask * entropy                            */
    ASE_ACC_CLIPPED,        /* sum mask * 1[|ratio-1| > e_clip]              */
    ASE_ACC_C_LOSS,         /* sum (return - value)^2                        */
    ASE_ACC_KL,             /* sum_rows policy_kl                            */
    ASE_ACC_DIV,            /* sum mask * diversity loss                     */
    ASE_ACC_BCE_AGENT,      /* sum softplus(l) over agent+replay rows        */
    ASE_ACC_BCE_DEMO,       /* sum softplus(-l) over demo rows               */
    ASE_ACC_AGENT_ACC,      /* count l < 0 (agent+replay)                    */
    ASE_ACC_DEMO_ACC,       /* count l > 0 (demo)                            */
    ASE_ACC_GP,             /* sum_rows |d logit / d x_demo|^2               */
    ASE_ACC_ENC,            /* sum_rows -<enc, z>                            */
    ASE_ACC_ENC_GP,         /* sum_rows |d enc_err / d x_agent|^2 (enc_grad_penalty); slots up to here are per-row sums
                               (the data-parallel ranks add them), the ones below depend on the weights only */
    ASE_ACC_LOGIT_W2,       /* sum w_logit^2                                 */
    ASE_ACC_DISC_W2,        /* sum over all disc weights^2                   */
    ASE_ACC_ENC_W2,         /* sum over all enc weights^2                    */
    ASE_ACC_GRAD_SQ,        /* sum of all gradients^2 (truncate_grads: global-norm clip) */
    ASE_ACC_COUNT = 24
};

/* acc[slot] += sum_i x[i] (f32 in, f64 atomic out); square != 0 sums x^2. */
int ase_hip_reduce_sum(const float* x, int64_t n, int square, double* acc, int slot, void* stream);

/* PPO actor/critic head for one minibatch (learning/common_agent.py:505-534,456-464,
 * learning/ase_agent.py:228-241,252-258,290-294,445-467; rl_games neglogp / policy_kl).
 *   mu      f32 [rows_mu, ld_mu]   rows [0,M) main pass, rows [M,2M) diversity pass (if div_on)
 *   value   f32 [M, ld_v] (column 0)
 *   mb_*    packed f32 minibatch fields (see ase_hip_gather_rows)
 *   d_mu    dtype [rows_mu, ld_dmu], d_value dtype [M, ld_dv]: d loss / d mu, d loss / d value
 *   db_mu f32[act_dim], db_value f32[1] (nullable): += their column sums (head bias gradients)
 *   masked: 1 -> sum(mask*x)/sum(mask) reductions (AMP/ASE), 0 -> plain means over m_global (PPO)
 *   mu_tanh: 1 -> mu_out = tanh(mu) precedes the losses (HRL high-level policy,
 *            learning/hrl_network_builder.py:26-29)
 *   acc[ASE_ACC_MASK_SUM] must already hold the GLOBAL mask sum.
 *   grad_scale (all loss heads): the STORED head gradients d_* are multiplied by it (the bias gradients and the loss
 *            scalars are not) - the static loss scale of ASE_F16 storage, whose back-propagated gradients would
 *            otherwise fall into half's subnormal range; the weight-gradient launches undo it through their alpha.
 *            The counterpart of the reference's GradScaler (learning/ase_agent.py:216,271-288).  1 for bf16 / f32.
 *            grad_scale_dev (nullable, ABI 6 / 7): a DEVICE scale record {factor on top - the DYNAMIC loss scale (ase_hip_scaler_step
 *            keeps it), read when the launch runs; overflow count, see ase_hip_gemm_nt's alpha_dev}.
 *   scratch: device f64[ASE_PPO_SCRATCH] workspace, private to one launch at a time: per-workgroup partial sums (loss
 *            scalars, head-bias column sums), folded into acc / db_* by a second one-workgroup kernel of the same call (no
 *            contended atomics; a kernel boundary instead of per-workgroup fences). */
#define ASE_PPO_SCRATCH (1024 * 72 + 8)
int ase_hip_ppo_head(const float* mu, int64_t ld_mu, const float* value, int64_t ld_v,
                     const float* mb_actions, const float* mb_old_mu, const float* mb_old_sigma,
                     const float* mb_old_logp, const float* mb_adv, const float* mb_old_value,
                     const float* mb_return, const float* mb_mask, const float* mb_z,
                     const float* new_z, const float* logstd,
                     void* d_mu, int64_t ld_dmu, void* d_value, int64_t ld_dv,
                     float* db_mu, float* db_value, float* mu_out, double* acc, double* scratch,
                     int M, int m_global, int act_dim, int z_dim, int masked, int div_on, int mu_tanh,
                     int clip_value, float e_clip, float critic_coef, float bounds_coef,
                     float div_coef, float div_tar, float grad_scale, float* grad_scale_dev, int dtype, void* stream);

/* Discriminator logit losses (learning/amp_agent.py:442-447,481-496): rows [0,2*amb) agent+replay
 * (target 0), rows [2*amb,3*amb) demo (target 1).  d_logit dtype [3*amb, ld_d] column 0. */
int ase_hip_disc_head(const float* logit, int64_t ld_l, void* d_logit, int64_t ld_d, float* db_logit,
                      double* acc, int amb, int amb_global, float disc_coef, float grad_scale, float* grad_scale_dev, int dtype,
                      void* stream);

/* Encoder head (learning/ase_network_builder.py:217, learning/ase_agent.py:413-418,469-472):
 * e f32 [amb, ld_e] pre-normalisation output, z f32 [amb, z_dim]; d_e dtype [amb, ld_de].
 * enc_out (nullable) f32 [amb, z_dim] receives normalize(e). */
int ase_hip_enc_head(const float* e, int64_t ld_e, const float* z, int64_t ld_z, void* d_e,
                     int64_t ld_de, float* db_enc, float* enc_out, double* acc, int amb, int amb_global,
                     int z_dim, float enc_coef, float grad_scale, float* grad_scale_dev, int dtype, void* stream);

/* Encoder gradient penalty (learning/ase_agent.py:431-441: mean_rows |d enc_err / d amp_obs|^2 with enc_err = -<normalize(e), z>),
 * the two per-row pieces around the GEMM chain.  e f32 [rows, ld_e] pre-normalisation encoder output, z f32 [rows, ld_z].
 *   seed: u[r, :] = scale * d enc_err / d e = -scale (z - eh <eh, z>) / |e|          (dtype [rows, ld_u]; eh = e / |e|)
 *   back: d_e[r, :] += grad_scale * J du[r, :],  J = d u / d e (unscaled), du f32 [rows, ld_du] = what the chain's backward
 *         returns at u (d_e carries the gradient scale of ase_hip_enc_head);
 *         db_enc (nullable, f32 [z_dim]) += column sums of the change of the stored d_e. */
int ase_hip_enc_gp_seed(const float* e, int64_t ld_e, const float* z, int64_t ld_z, void* u, int64_t ld_u, int rows,
                        int z_dim, float scale, int dtype, void* stream);
int ase_hip_enc_gp_back(const float* e, int64_t ld_e, const float* z, int64_t ld_z, const float* du, int64_t ld_du,
                        void* d_e, int64_t ld_de, float* db_enc, int rows, int z_dim, float grad_scale, float* grad_scale_dev,
                        int dtype, void* stream);

/* Gradient-penalty seed (learning/amp_agent.py:453-459): g[r,j] = scale * w[j] * act'  (d logit / d pre-activation of the last
 * hidden layer).  h = that layer's TWIN: its output for ReLU (act' = [h > 0]) and tanh (1 - h^2), its pre-activation z for
 * the smooth activations (act >= ASE_ACT_SILU). */
int ase_hip_gp_seed(const void* h, int64_t ld_h, const float* w, void* g, int64_t ld_g, int rows,
                    int width, float scale, int act, int dtype, void* stream);

/* Second-order term of the penalty's double backward for activations with curvature: with g = act'(z) u the chain value at a
 * layer and dg = act'(z) r the (masked) gradient of the penalty w.r.t. u,  dz[r,j] += act''(z) u r = act'' / act'^2 * g * dg
 * - the path through act'(z) that autograd's create_graph=True adds for anything but ReLU.  twin as in ase_hip_gp_seed. */
int ase_hip_gp_second(const void* twin, int64_t ld_t, const void* g, int64_t ld_g, const void* dg, int64_t ld_dg,
                      void* dz, int64_t ld_dz, int rows, int width, int act, int dtype, void* stream);

/* out[j] += scale * sum_r x[r, j] over an f32 matrix [rows, cols] (row pitch ld): the gradient of the logit weights from the
 * gradient penalty = column sums of the last launch of the chain's backward (learning/amp_agent.py:453-459) - a 3 us
 * stream instead of column sums inside that launch's store-bound epilogue. */
int ase_hip_colsum(const float* x, int64_t ld, int rows, int cols, float scale, float* out, void* stream);

/* acc[slot] += scale * sum_{r,j} x[r,j]^2 over a dtype matrix [rows, cols]; scale_dev (nullable): a DEVICE f32 factor on top. */
int ase_hip_sqnorm(const void* x, int64_t ld, int rows, int cols, double* acc, int slot, double scale, const float* scale_dev,
                   int dtype, void* stream);

/* train_result scalars from the accumulators (same keys as learning/ase_agent.py:296-306).
 * out f32[ASE_RES_COUNT].  opt_state_lr (nullable: constant lr): the optimizer state of ase_hip_begin_step - the learning rate
 * opt_state[1] is then adapted from this step's kl as rl_games' AdaptiveScheduler does under the 'legacy' schedule after every
 * minibatch (learning/common_agent.py:204-208; lr_schedule: adaptive, kl_threshold): kl > 2 thr: lr / 1.5 (>= 1e-6),
 * kl < thr / 2: lr * 1.5 (<= 1e-2) - on the device, no .item() per step.  out[ASE_RES_LR] = the learning rate THIS step was
 * taken with (the reference's train_result['last_lr'], learning/common_agent.py:430), 0 with a constant rate (the host
 * knows it).  ABI 3: ASE_RES_COUNT 16 -> 20 (ASE_RES_LR; result vectors are slots of a per-update ring, 80-byte pitch). */
enum {
    ASE_RES_A_LOSS = 0, ASE_RES_C_LOSS, ASE_RES_B_LOSS, ASE_RES_ENTROPY, ASE_RES_CLIP_FRAC, ASE_RES_KL,
    ASE_RES_DISC_LOSS, ASE_RES_DISC_GP, ASE_RES_DISC_LOGIT_LOSS, ASE_RES_DISC_AGENT_ACC,
    ASE_RES_DISC_DEMO_ACC, ASE_RES_ENC_LOSS, ASE_RES_DIV_LOSS, ASE_RES_LOSS, ASE_RES_MASK_SUM, ASE_RES_ENC_GP,
    ASE_RES_LR,
    ASE_RES_COUNT = 20
};
int ase_hip_finalize_scalars(const double* acc, float* out, int m_global, int amb_global, int masked,
                             int has_disc, int has_enc, int has_div, float critic_coef,
                             float entropy_coef, float bounds_coef, float disc_coef, float disc_logit_reg,
                             float disc_grad_penalty, float disc_weight_decay, float enc_coef,
                             float enc_weight_decay, float div_coef, float enc_grad_penalty, double* opt_state_lr,
                             float kl_threshold, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer (torch.optim.Adam(lr, eps=1e-8, weight_decay=0): learning/common_agent.py:45,
 * learning/ase_agent.py:265-269,287).
 * opt_state: DEVICE f64[8] = {step, lr, beta1, beta2, eps, bias_corr1, bias_corr2, _}; step is
 * advanced and the bias corrections recomputed on device by ase_hip_begin_step (graph-replay safe),
 * which also zeroes the n_acc accumulators.
 * ------------------------------------------------------------------------------------------- */
/* Global-norm gradient clipping, torch.nn.utils.clip_grad_norm_(parameters, grad_norm) (learning/ase_agent.py:273-288,
 * truncate_grads): g *= min(1, max_norm / (sqrt(*sqnorm) + 1e-6)); *sqnorm = sum of g^2 over all parameters (device f64,
 * e.g. accumulated by ase_hip_reduce_sum with square = 1). */
int ase_hip_clip_scale(float* g, int64_t n, const double* sqnorm, float max_norm, void* stream);

/* One small launch at the head of every optimisation step: + zeroes a second f64 buffer (the per-step partial statistics the
 * ranks exchange) and advances the Philox stream of the in-step latent draw (ase_hip_sample_latents with advance = 0). */
int ase_hip_begin_step(double* opt_state, double* acc, int n_acc, double* zero2, int n_zero2, uint64_t* rng_bump,
                       void* stream);
int ase_hip_adam(float* w, const float* g, float* m, float* v, int64_t n, const double* opt_state,
                 void* stream);
/* g[i] += c * w[i]  (the weight-only loss terms: learning/amp_agent.py:449-466) */
int ase_hip_axpy(float* g, const float* w, int64_t n, float c, void* stream);

/* Loss scaler of the half-storage engine = torch.cuda.amp.GradScaler around the optimizer step of the reference's mixed_precision
 * path (learning/ase_agent.py:271-288, learning/amp_agent.py:354-371: scaler.scale(loss).backward(); scaler.unscale_; scaler.step;
 * scaler.update - after EVERY optimisation step).  ABI 5; ABI 6: the scale lives on the device.
 *   scaler: DEVICE f64[8] = {found, skipped steps (total), growth tracker = clean steps since the scale last moved, steps (total),
 *           scale, growth_factor, backoff_factor, growth_interval}  (GradScaler's state and constructor arguments; the host writes
 *           slots 4-7 once)
 *   scale_tab: DEVICE f32[8] = four scale records {S, n0}, {1 / S, n1}, {1 / S^2, n2}, {1, n3}: what the `*_dev` arguments of the loss
 *           heads (S), the weight-gradient launches (1 / S), the penalty's norm (1 / S^2) and launches that carry no factor (1) point at
 *           - launches recorded once keep working when the scale moves - and where those launches report an overflow they stored (the
 *           counts n_k, ABI 7).
 * ase_hip_scaler_check = the found_inf test over ONE buffer the scaled backward wrote (dtype ASE_F32 / ASE_BF16 / ASE_F16):
 * scaler[found] += number of workgroups that met an element that is not finite or - ASE_F16, whose conversions saturate instead
 * of producing inf - sits at +-65504.  For buffers whose producers were given no record, and for the f32 gradient.
 * ase_hip_scaler_check_multi (ABI 7): the same over a DEVICE table int64[n_bufs][3] = {pointer, elements, dtype} in one launch,
 * wg_per_buf workgroups per buffer.
 * ase_hip_scaler_fold (ABI 7): scaler[found] += n0 + n1 + n2 + n3, counts = 0 - for a data-parallel step, whose ranks SUM-exchange
 * scaler[found] between this launch and ase_hip_scaler_step (which otherwise reads the counts itself). */
int ase_hip_scaler_check(const void* buf, int64_t n, int dtype, double* scaler, void* stream);
int ase_hip_scaler_check_multi(const int64_t* table, int n_bufs, int wg_per_buf, double* scaler, void* stream);
int ase_hip_scaler_fold(double* scaler, float* scale_tab, void* stream);
/* ase_hip_scaler_step = GradScaler.step + GradScaler.update, between the checks and the optimizer launch (ase_hip_adam reads opt_eff):
 *   found = scaler[found] != 0 or a non-zero count in scale_tab;
 *   found != 0: grads[0..n) = 0, opt_eff = the identity step {lr 0, beta1 = beta2 = 1, bias corrections 1} (w, m, v stay what they
 *               are), opt_state.step -= 1 (a skipped step is no optimizer step), skipped += 1;
 *               scale *= backoff_factor, growth tracker = 0
 *   found == 0: opt_eff = opt_state; growth tracker += 1, and when it reaches growth_interval: scale *= growth_factor, tracker = 0
 * then steps += 1, found = 0 and (scale_tab non-null) scale_tab = {scale, 0, 1 / scale, 0, 1 / scale^2, 0, 1, 0} - exactly
 * torch/amp/grad_scaler.py's _amp_update_scale_, once per optimisation step.  scale_tab NULL (ABI 5 behaviour): the scale is a
 * launch argument the host moves between updates; slots 4-7 are not touched. */
int ase_hip_scaler_step(double* scaler, double* opt_state, double* opt_eff, float* grads, int64_t n, float* scale_tab,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * Once-per-epoch rollout tail.
 * ------------------------------------------------------------------------------------------- */

/* r = -log(max(1 - sigmoid(l), 1e-4)) * scale   (learning/amp_agent.py:570-577) */
int ase_hip_disc_reward(const float* logit, int64_t ld_l, float* r, int64_t n, float scale, void* stream);
/* r = max(<normalize(e), z>, 0) * scale          (learning/ase_agent.py:404-411) */
int ase_hip_enc_reward(const float* e, int64_t ld_e, const float* z, int64_t ld_z, float* r, int64_t n,
                       int z_dim, float scale, void* stream);
/* GAE (learning/common_agent.py:437-449, learning/ase_agent.py:484-490,105-106):
 * rewards = w_task*task + w_disc*disc + w_enc*enc (disc/enc nullable); all [H,N] time-major. */
int ase_hip_gae(const uint8_t* dones, const float* values, const float* next_values,
                const float* r_task, const float* r_disc, const float* r_enc, float w_task, float w_disc,
                float w_enc, double gamma, double tau, float* advs, float* returns, int H, int N, void* stream);
/* advantages = returns - values, masked normalisation (learning/amp_agent.py:551-561 with rl_games
 * normalization_with_masks; mask nullable -> learning/common_agent.py:536-546).
 * Two calls: phase 0 accumulates moments into acc3 (f64[3], zeroed by the caller), phase 1 writes. */
int ase_hip_adv_norm(const float* returns, const float* values, const float* mask, float* adv,
                     double* acc3, int64_t n, int normalize, int phase, void* stream);

/* dst[(head + i) % size, :] = src[map(i), :]  (learning/replay_buffer.py:27-49) */
int ase_hip_ring_store(const float* src, int64_t ld_src, int D, const int32_t* idx, int remap_h,
                       int remap_n, int n, float* dst, int64_t size, int64_t head, void* stream);

/* y[r,:] = x[r,:] / max(|x[r,:]|_2, 1e-12): the latent a high-level action selects
 * (torch.nn.functional.normalize, learning/hrl_agent.py:235; learning/ase_network_builder.py:140-142). */
int ase_hip_normalize_rows(const float* x, int64_t ld_x, float* y, int64_t ld_y, int n, int dim, void* stream);

/* Rollout-time action head: the eval branch of the model wrapper + the eps-greedy mix
 * (learning/amp_models.py:29-36; learning/amp_agent.py:139-169; learning/ase_agent.py:117-148): per row
 * mu' = mu (tanh'ed when mu_tanh, learning/hrl_network_builder.py:26-29), sigma = exp(logstd), a = mu' + sigma N(0,1)
 * (Philox, rng_state as below), neglogp(a), rand_mask = Bernoulli(rand_probs[row]) and actions = rand_mask ? a : mu'.
 * rand_probs / rand_mask nullable (plain PPO: every row stochastic).  mu has leading dimension ld_mu, outputs are dense. */
int ase_hip_sample_actions(const float* mu, int64_t ld_mu, const float* logstd, const float* rand_probs,
                           uint64_t* rng_state, float* mu_out, float* sigma_out, float* actions, float* neglogp,
                           float* rand_mask, int n, int act_dim, int mu_tanh, void* stream);

/* z[r,:] = normalize(N(0,I))  (learning/ase_network_builder.py:221-225); counter-based Philox,
 * stream position read from and advanced in rng_state (u64[2] = {seed, offset}).  Row r draws the elements
 * (row_offset + r) * dim .. of the stream: data-parallel ranks pass the global index of their first row, so the R-rank
 * draw equals the 1-rank draw.  advance = 0: the stream position is left alone (ase_hip_begin_step moved it). */
int ase_hip_sample_latents(float* z, int rows, int dim, uint64_t* rng_state, int64_t row_offset, int advance,
                           void* z2 /* nullable: a second copy [rows, ld_z2] in z2_dtype (the GEMM input buffer) */,
                           int64_t ld_z2, int z2_dtype, void* stream);

/* The whole optimizer step of every dense layer in ONE launch: weight-only gradient terms (g += c * w: discriminator
 * weight decay / logit regulariser / encoder weight decay), their reported sums of squares (pre-update weights, into
 * acc[slot]), torch.optim.Adam, and the refreshed compute-dtype shadows W_s / W_s^T / bias (ase_hip_refresh_shadow_multi's
 * job).  desc: DEVICE int64[n_layers][24] = {W, n_real, k_real, Ws, ldws, Wts, ldwts, split_src, split_dst - split_src,
 * bias, bias_shadow, ceil(k_real/32), gW, mW, vW, gb, mb, vb, coefficient c (f32 bit pattern), acc slot A or -1,
 * acc slot B or -1, 0, 0, 0}.  opt_state NULL: shadows only.
 * Replaces: the weight terms of learning/amp_agent.py:449-466 + optimizer.step() (learning/ase_agent.py:287). */
int ase_hip_apply_multi(const int64_t* desc, int n_layers, const double* opt_state, double* acc, int dtype,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * Observation side.
 * ------------------------------------------------------------------------------------------- */

/* One frame of the AMP (discriminator) observation per environment from the simulator state, pushed into the history
 * hist [n_envs, n_steps, F], F = 13 + 6 n_joints + n_dof + 3 n_key: {root height, root rotation as tangent + normal
 * (heading-local when local_root_obs), heading-local root velocity and angular velocity, per joint tangent + normal of
 * its rotation (3-dof joints: exponential map; 1-dof: hinge about y), dof velocities, heading-local key body
 * positions}.  shift != 0: slots move one step into the past first; the new frame takes slot 0.  Quaternions xyzw.
 * dof_offsets: HOST int32[n_joints + 1].
 * Replaces: build_amp_observations + _update_hist_amp_obs (env/tasks/humanoid_amp.py:248-266,280-316),
 *   dof_to_obs (env/tasks/humanoid.py:523-552). */
int ase_hip_build_amp_obs(const float* root_pos, const float* root_rot, const float* root_vel,
                          const float* root_ang_vel, const float* dof_pos, const float* dof_vel,
                          const float* key_body_pos, int n_envs, int n_dof, int n_key, const int32_t* dof_offsets,
                          int n_joints, int local_root_obs, int root_height_obs, float* hist, int n_steps, int shift,
                          void* stream);

/* Motion-clip sampler: the reference pose at (motion id, time) for n samples from the concatenated clip tensors
 * (global translations gts [frames, B, 3], global / local rotations grs / lrs [frames, B, 4] xyzw, root velocities
 * grvs / gravs [frames, 3], dof velocities dvs [frames, D]; per motion: lengths [s], num_frames, dt, length_starts):
 * frame pair + blend, linear interpolation of positions, spherical interpolation of rotations, local rotations ->
 * dof positions (3-dof: exponential map, 1-dof: angle about y).  Outputs feed ase_hip_build_amp_obs (demo stream).
 * dof_body_ids / dof_offsets / key_body_ids: HOST int32 arrays; everything else device memory.
 * Replaces: MotionLib.get_motion_state (utils/motion_lib.py:122-172,263-272,296-325). */
int ase_hip_motion_state(const float* gts, const float* grs, const float* lrs, const float* grvs, const float* gravs,
                         const float* dvs, int n_bodies, const float* lengths, const int32_t* num_frames, const float* dt,
                         const int32_t* length_starts, const int32_t* motion_ids, const float* times, int n,
                         const int32_t* dof_body_ids, const int32_t* dof_offsets, int n_joints,
                         const int32_t* key_body_ids, int n_key, float* root_pos, float* root_rot, float* dof_pos,
                         float* root_vel, float* root_ang_vel, float* dof_vel, float* key_pos, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Launch programs: record a sequence of the calls above ONCE (nothing is launched while recording), replay it with 4-5 us
 * of host work per launch on the same HIP streams - the optimisation step as one call, with OUR branch -> stream mapping
 * (a captured hipGraph picks its own; eager launches from Python fall behind the GPU).  Recording is per thread.
 * ase_hip_mark / ase_hip_wait are the fork / join points between streams (event record / stream-wait-event); outside a
 * recording they act immediately on a pool of events.  ase_hip_memset / ase_hip_memcpy: recordable fills / device copies.
 * ------------------------------------------------------------------------------------------- */
int ase_hip_prog_create(void** prog);
int ase_hip_prog_destroy(void* prog);
int ase_hip_prog_begin(void* prog);
int ase_hip_prog_end(void* prog);
int ase_hip_prog_size(void* prog);          /* entries recorded (launches + fork / join points); < 0: null program */
int ase_hip_prog_launch(void* prog);
/* A HOST callback at this position of the sequence: called immediately outside a recording, on every replay inside one.
 * This is how the exchange points of the data-parallel update (the RCCL all-reduce of a branch's gradient bucket, issued while
 * the other branches' backward launches are still being replayed - learning/common_agent.py:94-107, the Horovod optimizer's
 * gradient hooks) live inside ONE launch program; fn runs on the replaying thread and must enqueue its work on a stream
 * itself. */
int ase_hip_prog_host(void (*fn)(void*), void* arg);
int ase_hip_mark(void* stream, int* id);
int ase_hip_wait(void* stream, int id);
int ase_hip_memset(void* dst, int value, int64_t bytes, void* stream);
int ase_hip_memcpy(void* dst, const void* src, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ASE_HIP_H */
