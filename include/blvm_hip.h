/* blvm_hip.h — C ABI of libblvm_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the blvm hot path.
 *
 * The reference (JakobHavtorn/benchmarking-lvms, package `blvm`) is pure Python on PyTorch and has NO FFI of its
 * own (`setup.py:57` ext_modules=[]); every entry point below replaces a chain of ATen ops at the cited reference
 * lines (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - All tensors are device pointers to row-major fp32 unless stated; reductions that the reference carries in
 *     float64 (`blvm/models/vrnn.py:266`) are double.
 *   - The caller owns every buffer (inputs, outputs, workspace/reserve); the library never allocates, frees or
 *     retains a pointer past the call.
 *   - All work is enqueued on the caller's `stream` (a hipStream_t passed as void*); no implicit device sync.
 *   - Return value: 0 on success, a negative BLVM_E* code otherwise; text via blvm_last_error() (thread-local).
 *   - Sequence tensors are TIME-MAJOR [T', B, F] (one contiguous slab per recurrent step).
 */
#ifndef BLVM_HIP_H
#define BLVM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLVM_OK 0
#define BLVM_EINVAL (-1)  /* bad shape / alignment / null pointer */
#define BLVM_ELAUNCH (-2) /* HIP launch or runtime error */
#define BLVM_ENOSUP (-3)  /* configuration not supported by this build */

int blvm_version(void);
const char* blvm_last_error(void);
/* 1 if a gfx950 device is visible to the HIP runtime, else 0 (never throws). */
int blvm_device_ok(void);
/* The recurrent sequences (K1-K5) run as ONE persistent launch whose workgroups wait for each other with BOUNDED spins.  A spin
 * that gives up (a workgroup of the launch was not resident: another process holds CUs) drains its launch with garbage results; the
 * call that enqueued it has long returned BLVM_OK.  This reports such launches: the number so far in this process (0 = none) and
 * the code (step << 4 | link) of the last spin that failed.  Reads pinned host memory; never blocks.  Check it where results are
 * read back (the reference's loops synchronise when they read the loss, `experiments/experiment_vrnn_audio.py:232`). */
int blvm_async_errors(unsigned* last_code);
/* The same counter with read-and-clear semantics: aborted launches since the previous take (and the code of the last one, 0 if
 * none).  For loops that handle an abort (drop the step, all-reduce the count over ranks so that every rank leaves together) and go
 * on: the sticky total above would fail every later check of the process. */
int blvm_async_errors_take(unsigned* last_code);
/* Execution switch of K1-K5 (results agree to fp32 summation order): sequences with at most `max_batch` rows run as one
 * persistent launch, larger ones as one launch per link (0 = always per link; < 0 = leave unchanged; default 128 or env
 * BLVM_PCHAIN_MAX_B / BLVM_PCHAIN=0; the VRNN pair also runs batches of 65 .. 256 rows persistently, on 32-row tiles).  `waves` = 8 or
 * 16 waves per workgroup of the persistent kernels (other values: unchanged; default 16, env BLVM_PCHAIN_NW). */
int blvm_pchain_configure(int max_batch, int waves);
int blvm_pchain_max_batch(void); /* the current limit (at most 128) */
/* Operand type of the matrix products — the reference's `--use_amp True` switch (`experiments/experiment_vrnn_audio.py:198,219-230`:
 * torch.autocast around forward).  BLVM_DTYPE_F32 (default): fp32 operands, exact fp32 fma chains.  BLVM_DTYPE_BF16: the
 * persistent recurrent chains (K1-K5 with B <= blvm_pchain_max_batch()) and the K6 GEMMs round their operands to bf16 (nearest
 * even) and multiply on the bf16 matrix pipe; accumulation, epilogues, reductions, likelihoods and everything stored stay fp32
 * (autocast keeps bf16 outputs; this mode is at least as precise).  Also settable with env BLVM_DTYPE=bf16 before the first call. */
#define BLVM_DTYPE_F32 0
#define BLVM_DTYPE_BF16 1
int blvm_set_operand_dtype(int dtype);
int blvm_get_operand_dtype(void);
/* Diagnostics: while `device_buffer` (64 zero-initialised uint64 in device memory, caller-owned) is installed, the persistent kernels
 * add the 100 MHz wall-clock ticks two of their workgroups spend in every descriptor of the step program (waits included): words
 * [0..31] workgroup 0 (critical path), [32..55] the first workgroup of the deferred range ([64..127]: the same for backward
 * programs; pass 128 words).  NULL uninstalls. */
int blvm_pchain_profile(unsigned long long* device_buffer);
/* Diagnostics: placement bits of the persistent programs (results unchanged; default 20): 4 XCD-aware column-tile placement (a
 * column tile's row tiles on one XCD, so every XCD's L2 holds 1/8 of each weight matrix), 16 one-word canary poll in front of the
 * operand polls of tiles off the critical path. */
int blvm_pchain_tune(int bits);
/* Diagnostics / unit test of the persistent-chain engine on its own: L dependent links x_{s+1} = relu(x_s W^T + b), [B,N] x [N,N],
 * as ONE launch.  W16: W [N,N] in the T16 operand layout (blvm_pchain_rows_to_t16 of W: a weight's rows are the "batch");
 * x16: L+1 T16 slabs of ceil(B/16)*16 x N floats, slab 0 = x_0 in T16 (blvm_pchain_rows_to_t16); xs: L row-major [B,N] outputs.
 * nwg: workgroups (0: one per tile, at most one per CU). */
int blvm_pchain_chain_probe(const float* W16, const float* bias, float* x16, float* xs, int B, int N, int L, int nwg, void* stream);
/* dst = T16 operand copy [ceil(B/16)*16, K] of the rows of src [B,K] (row stride ld). */
int blvm_pchain_rows_to_t16(const float* src, int ld, int B, int K, float* dst, void* stream);
/* n host integers -> device memory through kernel arguments (asynchronous on `stream`; a pageable hipMemcpy would block the host
 * until the stream has drained).  Carries the batch's lengths `x_sl` (`blvm/data/batchers.py:145-151` hands them over on the host). */
int blvm_upload_i32(const int32_t* host, int n, int32_t* dst, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K6  fp32 MFMA GEMM with fused epilogue — replaces nn.Linear(+LeakyReLU/ReLU) chains
 *     (`blvm/models/vrnn.py:487-505` encoder/decoder MLPs; autograd of the same for the backward forms).
 *
 *   C[M,N] (=|+=) epi( sum_k A'(m,k) * B'(k,n) )
 *     op_a = 0: A'(m,k) = A[m*lda + k]      op_a = 1: A'(m,k) = A[k*lda + m]
 *     op_b = 0: B'(k,n) = B[n*ldb + k]      op_b = 1: B'(k,n) = B[k*ldb + n]
 *   epilogue (in this order): + bias[n] (if bias) ; activation act (0 none, 1 ReLU, 2 LeakyReLU(slope)) ;
 *     * dact(gate[m*ldg+n]) (if gate: 1 where gate>0 else slope — derivative of the activation whose OUTPUT is gate)
 *   accumulate != 0: C += result (split_k > 1 forces atomic accumulation; caller zeroes C unless accumulating).
 * ------------------------------------------------------------------------------------------------------------- */
int blvm_gemm_f32(int op_a, int op_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                  float* C, int ldc, const float* bias, int act, float slope, const float* gate, int ldg,
                  int accumulate, int split_k, void* stream);

/* dz[i] = dy[i] * (y[i] > 0 ? 1 : slope) over n contiguous floats: derivative of ReLU / LeakyReLU from its OUTPUT y
 * (used where no producing GEMM exists to fuse it into, e.g. behind the DMoL head).  dz may alias dy. */
int blvm_act_bwd_f32(const float* dy, const float* y, float slope, float* dz, size_t n, void* stream);

/* Weight + bias gradient of y = x W^T + b in ONE launch (autograd's `grad_weight = dy^T x`, `grad_bias = dy.sum(0)`):
 *   dW[n*lddw + k] += sum_r D[r*ldd + n] X[r*ldx + k]   ;   db[n] += sum_r D[r*ldd + n]     (both ACCUMULATE; caller zeroes)
 * D [rows, N_out] pre-activation gradients, X [rows, K_in] the layer's input; dW or db may be NULL; split_k < 1: chosen here.
 * The workgroups of the GEMM's first column block sum the D tiles they stage anyway (fp32 operands; in the bf16-operand mode the
 * bias gradient stays an fp32 sum in its own launch). */
int blvm_wgrad_f32(int N_out, int K_in, int rows, const float* D, int ldd, const float* X, int ldx, float* dW, int lddw, float* db,
                   int split_k, void* stream);

/* n weight (+ bias) gradients over the SAME rows as ONE launch: job i is blvm_wgrad_f32(N_out[i], K_in[i], rows, D[i], ldd[i], X[i],
 * ldx[i], dW[i], lddw[i], db[i]) — the layers of one MLP chain, the links of one recurrent sequence (reference: autograd's per-layer
 * `grad_weight = dy^T x`, torch/nn/functional.linear backward, called once per nn.Linear of blvm/models/vrnn.py:60-75).  Small
 * outputs (256 x 256) cannot fill the chip alone without a split so fine that the atomics dominate; together they share one coarse
 * split.  dW[i] / db[i] may be NULL.  Falls back to one launch per job for bf16 operands, rows < 1024 or more than 20 jobs. */
int blvm_wgrad_group_f32(int n, const int* N_out, const int* K_in, int rows, const float* const* D, const int* ldd, const float* const* X,
                         const int* ldx, float* const* dW, const int* lddw, float* const* db, void* stream);

/* out[n] (=|+=) sum_m X[m*ldx + n]   (bias gradients). */
int blvm_colsum_f32(int M, int N, const float* X, int ldx, float* out, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K7  fused DMoL head: Linear(30->30) + split/clamp + discretized-logistic-mixture log-likelihood + mask + per-
 *     utterance sum.  Replaces `blvm/modules/distributions.py:381-387` + `blvm/utils/log_likelihoods.py:170-231`
 *     + the masked sum at `blvm/models/vrnn.py:266-269`.
 *
 *   dec      [rows, S*F] contiguous, F = 3*num_mix features per audio frame (frame f = row*S + j at dec + f*F)
 *   layout   0: rows are batch-major (row = b*Tp + t)   1: rows are time-major (row = t*B + b)
 *   y        [B, T] targets in [-1,1];  x_sl [B] int32 on device;  frame (b, tau=t*S+j) counts iff tau < x_sl[b]
 *   W [F,F], bias [F]: the likelihood's Linear (both NULL: `dec` already holds the F parameters, e.g. WaveNet's
 *   Linear(C->F) computed by K6);  log_eps: clamp floor of the log-scales (-7)
 *   log_prob [B] double, ACCUMULATED (caller zeroes);  ll_twise optional [B,T] fp32 (masked ll, may be NULL)
 * ------------------------------------------------------------------------------------------------------------- */
int blvm_dmol_fwd(const float* dec, int layout, const float* W, const float* bias, const float* y,
                  const int32_t* x_sl, int B, int T, int Tp, int S, int num_mix, int num_bins, float log_eps,
                  double* log_prob, float* ll_twise, void* stream);

/*   g_b [B] fp32 = dLoss/dlog_prob[b].  Writes d_dec (same shape as dec; zero on masked frames) and d_par
 *   [rows*S, F] = gradient wrt the Linear's OUTPUT (for dW = d_par^T dec, db = colsum(d_par)). */
int blvm_dmol_bwd(const float* dec, int layout, const float* W, const float* bias, const float* y,
                  const int32_t* x_sl, const float* g_b, int B, int T, int Tp, int S, int num_mix, int num_bins,
                  float log_eps, float* d_dec, float* d_par, void* stream);

/* K7b / K7c  Gaussian output heads, same calling convention, frame layout, masks and float64 per-utterance sums as K7:
 *   gmm: `DiagonalGaussianMixtureDense.forward` + `.log_prob` (`blvm/modules/distributions.py:153-206`,
 *        `gaussian_mixture_ll` `blvm/utils/log_likelihoods.py:42-60`): dec [rows, S*3*num_mix] -> Linear(30->30) (W NULL:
 *        dec ARE the parameters) -> logits | means | pre-softplus sds, sd = softplus_beta(raw) + sd_eps.
 *   gauss_head: `DiagonalGaussianDense.forward` + `.log_prob` (`distributions.py:105-150`, `gaussian_ll`
 *        `log_likelihoods.py:17-39`): dec [rows, S*2] -> Linear(2->2) (W NULL: none) -> mean | pre-softplus sd.
 *   bwd: d_dec (gradient wrt dec) and d_par (gradient wrt the Linear's output; dW = d_par^T dec, db = colsum(d_par)). */
int blvm_gmm_fwd(const float* dec, int layout, const float* W, const float* bias, const float* y, const int32_t* x_sl,
                 int B, int T, int Tp, int S, int num_mix, float sd_beta, float sd_eps, double* log_prob, float* ll_twise,
                 void* stream);
int blvm_gmm_bwd(const float* dec, int layout, const float* W, const float* bias, const float* y, const int32_t* x_sl,
                 const float* g_b, int B, int T, int Tp, int S, int num_mix, float sd_beta, float sd_eps, float* d_dec,
                 float* d_par, void* stream);
int blvm_gauss_head_fwd(const float* dec, int layout, const float* W, const float* bias, const float* y,
                        const int32_t* x_sl, int B, int T, int Tp, int S, float sd_beta, float sd_eps, double* log_prob,
                        float* ll_twise, void* stream);
int blvm_gauss_head_bwd(const float* dec, int layout, const float* W, const float* bias, const float* y,
                        const int32_t* x_sl, const float* g_b, int B, int T, int Tp, int S, float sd_beta, float sd_eps,
                        float* d_dec, float* d_par, void* stream);

/* Samplers / modes of the mixture heads (`DiscretizedLogisticMixtureDense.sample/.mode`, `DiagonalGaussianMixtureDense.sample/.mode`,
 * `blvm/modules/distributions.py:173-186,359-368`; `blvm/utils/variational.py:156-195,283-349`).  par [n, 3*num_mix] = head outputs
 * (logits | locations | raw scales); u [n,num_mix] uniforms for the Gumbel-max component pick (NULL: arg-max logit = mode);
 * v [n] the picked component's noise (kind 0 DMoL: uniform -> logistic, clamped to [-1,1]; kind 1 GMM: standard normal;
 * NULL: the location).  out [n].  The caller draws u, v (RNG stays with the host framework, as for eps). */
int blvm_mix_sample(const float* par, const float* u, const float* v, long long n, int num_mix, int kind, float log_eps,
                    float sd_beta, float sd_eps, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K8  fused analytic Gaussian KL + free-nats + mask + per-utterance sums.  Replaces
 *     `blvm/utils/variational.py:67-70,86-122` + `blvm/models/vrnn.py:271-276`.
 *   mu_q, sd_q, mu_p, sd_p [rows=Tp*B, Z] (layout as above); step t of row b counts iff t*stride < x_sl[b]
 *   fn_floor = free_nats / Z (<= 0 disables);  kld, kld_fn [B] double, ACCUMULATED (caller zeroes)
 * ------------------------------------------------------------------------------------------------------------- */
int blvm_kl_fwd(const float* mu_q, const float* sd_q, const float* mu_p, const float* sd_p, int layout,
                const int32_t* x_sl, int B, int Tp, int Z, int stride, float fn_floor, double* kld,
                double* kld_fn, void* stream);

/*   c_raw[b] = dLoss/dkld[b], c_fn[b] = dLoss/dkld_fn[b] (fp32, either may be NULL = 0). */
int blvm_kl_bwd(const float* mu_q, const float* sd_q, const float* mu_p, const float* sd_p, int layout,
                const int32_t* x_sl, const float* c_raw, const float* c_fn, int B, int Tp, int Z, int stride,
                float fn_floor, float* d_mu_q, float* d_sd_q, float* d_mu_p, float* d_sd_p, void* stream);

/* K8b  Gaussian latent head, elementwise over n = rows*Z: sd = softplus_beta(raw) + sd_eps for prior and posterior
 * (`DiagonalGaussianDenseSTCN.forward`, `blvm/models/stcn/stcn.py:32-76`), posterior combination (mode 0 plain, 1 residual
 * mu_q += mu_p, 2 precision-weighted `variational.py:125-138`) and the reparameterised sample z = mu_q + sd_q*eps
 * (`variational.py:141-152`) — the per-level body of `STCN.infer` (`stcn.py:297-327`).  bwd: upstream gradients wrt the four
 * outputs (each may be NULL) -> gradients wrt mu_p, sd_p_raw, mu_q_raw, sd_q_raw. */
int blvm_gauss_latent_fwd(const float* mu_p, const float* sd_p_raw, const float* mu_q_raw, const float* sd_q_raw,
                          const float* eps, size_t n, float beta_p, float beta_q, float sd_eps, int mode, float* sd_p,
                          float* mu_q, float* sd_q, float* z, void* stream);
int blvm_gauss_latent_bwd(const float* mu_p, const float* sd_p_raw, const float* mu_q_raw, const float* sd_q_raw,
                          const float* eps, const float* g_sd_p, const float* g_mu_q, const float* g_sd_q, const float* g_z,
                          size_t n, float beta_p, float beta_q, float sd_eps, int mode, float* d_mu_p, float* d_sd_p_raw,
                          float* d_mu_q_raw, float* d_sd_q_raw, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K1  VRNN recurrent cell over a whole sequence (forward and BPTT).  Replaces the scripted per-step loop
 *     `blvm/models/vrnn.py:305-308` over `VRNNCell.forward` (`vrnn.py:109-141`) and its autograd backward.
 *     condition_h_on_x = True (the only form VRNNAudio builds, `vrnn.py:506-516`).
 *
 *   weights: reference layout, Linear [out,in], GRU rows [r|z|n] (`SURVEY.md` §8b).
 *   X = x_dim (encoder feature size), H = h_dim (MLP width), Z = z_dim, R = r_dim (GRU state).
 *   All of X, H, Z, R must be multiples of 16; B is arbitrary (row-guarded tiles).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct BlvmVrnnWeights {
  const float *prior_w[3], *prior_b[3]; /* [H,R], [H,H], [H,H] */
  const float *prior_hw, *prior_hb;     /* [2Z,H] */
  const float *post_w[3], *post_b[3];   /* [H,R+X] (input order cat[h,x]), [H,H], [H,H] */
  const float *post_hw, *post_hb;       /* [2Z,H] */
  const float *phi_w[4], *phi_b[4];     /* [H,Z], [H,H] x3 */
  const float *gru_wih, *gru_whh;       /* [3R, X+H] (input order cat[x,phi]), [3R,R] */
  const float *gru_bih, *gru_bhh;       /* [3R] */
} BlvmVrnnWeights;

typedef struct BlvmVrnnGrads { /* same shapes as the weights; ACCUMULATED into (caller zeroes) */
  float *prior_w[3], *prior_b[3], *prior_hw, *prior_hb;
  float *post_w[3], *post_b[3], *post_hw, *post_hb;
  float *phi_w[4], *phi_b[4];
  float *gru_wih, *gru_whh, *gru_bih, *gru_bhh;
} BlvmVrnnGrads;

/* K1c  Ancestral sampling from VRNNAudio: every step of every utterance in ONE launch.  Replaces the loop of `VRNN.generate`
 *   (`blvm/models/vrnn.py:371-434`): enc = encoder(x_t) -> `VRNNCell.generate` (`vrnn.py:143-164`: prior -> z = mu + sd eps ->
 *   phi_z -> GRU) -> decoder(cat[phi_z, h_new]) -> DMoL head per sample -> draw -> x_{t+1}.  All weights in their PyTorch layouts;
 *   the encoder / decoder are 3 x (Linear + LeakyReLU(slope)) as in `vrnn.py:487-505`, x_dim = H.
 *   x0 [B,S] first frame stack; h0 [B,R] or NULL (zeros); eps [T,B,Z] standard normal; u [T,B,S,num_mix], v [T,B,S] uniforms of the
 *   DMoL sampler as in blvm_mix_sample (both NULL: the mode); x_out [B,T,S]; h_out [B,R] or NULL.
 *   scratch: blvm_vrnn_decode_scratch_floats(...) floats (operand-layout copies of the weights). */
typedef struct BlvmVrnnDecodeWeights {
  const float *enc_w[3], *enc_b[3]; /* [H,S], [H,H], [H,H] */
  const BlvmVrnnWeights* cell;      /* prior_*, phi_*, gru_* are read (the posterior is not evaluated when z comes from the prior) */
  const float *dec_w[3], *dec_b[3]; /* [H,H+R] (input order cat[phi_z, h]), [H,H], [S*3*num_mix,H] */
  const float *lik_w, *lik_b;       /* [3*num_mix, 3*num_mix], [3*num_mix]: the head's Linear, applied per sample */
} BlvmVrnnDecodeWeights;
size_t blvm_vrnn_decode_scratch_floats(int S, int H, int Z, int R);
int blvm_vrnn_decode(const BlvmVrnnDecodeWeights* w, const float* x0, const float* h0, const float* eps, const float* u,
                     const float* v, int T, int B, int S, int H, int Z, int R, int num_mix, float sd_eps, float slope,
                     float log_eps, float* x_out, float* h_out, float* scratch, void* stream);

/* K1c on the whole chip: the same sampling loop (same arguments and draws, B <= 128) as ONE persistent launch whose every layer is a
 * link dealt over all CUs (csrc/pchain.hip) instead of 16 utterances per CU: 17 links per step.  scratch:
 * blvm_vrnn_generate_scratch_floats(T, B, ...) floats (weight copies + one slab per step of every activation). */
size_t blvm_vrnn_generate_scratch_floats(int T, int B, int S, int H, int Z, int R);
int blvm_vrnn_generate(const BlvmVrnnDecodeWeights* w, const float* x0, const float* h0, const float* eps, const float* u,
                       const float* v, int T, int B, int S, int H, int Z, int R, int num_mix, float sd_eps, float slope,
                       float log_eps, float* x_out, float* h_out, float* scratch, void* stream);

/* number of floats of `reserve` (activations kept for BPTT) / `workspace` (backward scratch). */
size_t blvm_vrnn_reserve_floats(int Tp, int B, int X, int H, int Z, int R);
size_t blvm_vrnn_bwd_workspace_floats(int Tp, int B, int X, int H, int Z, int R);

/*   enc [Tp,B,X]; h0 [B,R] (NULL = zeros); eps [Tp,B,Z] standard-normal noise (reference draws it per step,
 *   `blvm/utils/variational.py:141-152`); sd_eps = epsilon of the Gaussian heads (1e-6), initial_sd = 1.
 *   decin [Tp+1,B,H+R]: row t = [phi_t | h_{t-1}] — the decoder input `cat([phi_z, h])` of `vrnn.py:321-324`;
 *     row Tp carries the final state in its h-part.
 *   mu_q (residual already added), sd_q, mu_p, sd_p, z: [Tp,B,Z]. */
int blvm_vrnn_seq_fwd(const BlvmVrnnWeights* w, const float* enc, const float* h0, const float* eps, int Tp, int B,
                      int X, int H, int Z, int R, int residual_posterior, float sd_eps, float* decin,
                      float* mu_q, float* sd_q, float* mu_p, float* sd_p, float* z, float* reserve, void* stream);

/*   d_decin [Tp+1,B,H+R]: gradient wrt decin (row Tp ignored).  KL term folded in: loss += sum_b (c_raw[b] *
 *   kld[b] + c_fn[b] * kld_fn[b]) with the mask t*stride < x_sl[b] and floor fn_floor (see blvm_kl_fwd);
 *   c_raw / c_fn may be NULL (= 0).
 *   Outputs: d_enc [Tp,B,X], d_h0 [B,R] (may be NULL), grads (accumulated). */
int blvm_vrnn_seq_bwd(const BlvmVrnnWeights* w, const float* enc, const float* eps, const float* decin,
                      const float* mu_q, const float* sd_q, const float* mu_p, const float* sd_p, const float* z,
                      const float* reserve, const float* d_decin, const int32_t* x_sl, const float* c_raw,
                      const float* c_fn, int stride, float fn_floor, int Tp, int B, int X, int H, int Z, int R,
                      int residual_posterior, float sd_eps, float* d_enc, float* d_h0, const BlvmVrnnGrads* grads,
                      float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K4  single-layer LSTM over a sequence (forward + BPTT).  Replaces `nn.LSTM` on a packed sequence,
 *     `blvm/models/lstm.py:46-55,96-98` (pack_padded_sequence / pad_packed_sequence semantics through `lens`:
 *     a row past its length keeps its state and emits zeros).  Gate rows [i|f|g|o].
 *   in [T,B,I]; h0,c0 [B,H] or NULL (zeros); lens [B] int32 valid steps per row, or NULL (all T)
 *   out [T,B,H]; hn,cn [B,H] state after each row's last valid step (may be NULL)
 * ------------------------------------------------------------------------------------------------------------- */
size_t blvm_lstm_reserve_floats(int T, int B, int H);
size_t blvm_lstm_bwd_workspace_floats(int T, int B, int H);
int blvm_lstm_seq_fwd(const float* Wih, const float* Whh, const float* bih, const float* bhh, const float* in,
                      const float* h0, const float* c0, const int32_t* lens, int T, int B, int I, int H, float* out,
                      float* hn, float* cn, float* reserve, void* stream);
/*   d_out [T,B,H]; outputs d_in [T,B,I], d_h0, d_c0 [B,H] (each may be NULL); weight grads ACCUMULATED (may be NULL). */
int blvm_lstm_seq_bwd(const float* Wih, const float* Whh, const float* in, const float* reserve, const float* d_out,
                      int T, int B, int I, int H, float* d_in, float* d_h0, float* d_c0, float* dWih, float* dWhh,
                      float* dbih, float* dbhh, float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K2  single-layer GRU over a sequence (forward + BPTT), optionally time-reversed PER ROW.  Replaces `nn.GRU` and
 *     the `reverse_sequences` gathers around it, `blvm/models/srnn.py:113-116,196,200-206` +
 *     `blvm/utils/operations.py:56-87`: with reverse != 0 row b consumes time index lens[b]-1-j at recurrence step
 *     j < lens[b] and index j in its right padding, and its outputs are written back at those time indices.
 *     Gate rows [r|z|n] (torch.nn.GRU).
 *   in [T,B,I] with row stride ld_in; h0 [B,R] or NULL; out element (t,b,c) at out + t*out_ts + b*out_ld + c, so the
 *   states can be written straight into a wider time-major buffer; hn [B,R] state after the last recurrence step.
 * ------------------------------------------------------------------------------------------------------------- */
size_t blvm_gru_reserve_floats(int T, int B, int R);
size_t blvm_gru_bwd_workspace_floats(int T, int B, int R);
int blvm_gru_seq_fwd(const float* Wih, const float* Whh, const float* bih, const float* bhh, const float* in,
                     int ld_in, const float* h0, const int32_t* lens, int reverse, int T, int B, int I, int R,
                     float* out, long long out_ts, int out_ld, float* hn, float* reserve, void* stream);
/*   d_out addressed like out.  d_in [T,B,I] row stride ld_din (=|+= by accumulate_din), d_h0 [B,R]; weight grads
 *   ACCUMULATED; any output may be NULL. */
int blvm_gru_seq_bwd(const float* Wih, const float* Whh, const float* in, int ld_in, const int32_t* lens, int reverse,
                     const float* reserve, const float* d_out, long long out_ts, int out_ld, int T, int B, int I,
                     int R, float* d_in, int ld_din, int accumulate_din, float* d_h0, float* dWih, float* dWhh,
                     float* dbih, float* dbhh, float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K3  SRNN latent chain over a sequence (forward + BPTT).  Replaces the Python loop `blvm/models/srnn.py:224-253`
 *     (prior / posterior MLPs `srnn.py:92-111` on cat[d_t, z_{t-1}] / cat[a_t, z_{t-1}], residual posterior, rsample)
 *     and its autograd backward; the non-gated ("Elman") stochastic transfer that SRNNAudio builds.
 *   d, a [Tp,B,R]: deterministic forward state (shifted) and smoothing state; eps [Tp,B,Z]; z0 [B,Z] or NULL.
 *   zs [Tp+1,B,Z]: row 0 = z0, row t+1 = z_t.  slope: LeakyReLU slope of the MLPs (0.01).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct BlvmSrnnWeights {
  const float *prior_w[3], *prior_b[3]; /* [H,R+Z] (input order cat[d,z]), [H,H], [H,H] */
  const float *prior_hw, *prior_hb;     /* [2Z,H] */
  const float *post_w[3], *post_b[3];   /* [H,R+Z] (input order cat[a,z]), [H,H], [H,H] */
  const float *post_hw, *post_hb;       /* [2Z,H] */
} BlvmSrnnWeights;

typedef struct BlvmSrnnGrads { /* same shapes; ACCUMULATED into (caller zeroes) */
  float *prior_w[3], *prior_b[3], *prior_hw, *prior_hb;
  float *post_w[3], *post_b[3], *post_hw, *post_hb;
} BlvmSrnnGrads;

size_t blvm_srnn_reserve_floats(int Tp, int B, int H, int Z, int R);
size_t blvm_srnn_bwd_workspace_floats(int Tp, int B, int H, int Z, int R);
int blvm_srnn_latent_fwd(const BlvmSrnnWeights* w, const float* d, const float* a, const float* z0, const float* eps,
                         int Tp, int B, int H, int Z, int R, int residual_posterior, float sd_eps, float slope,
                         float* zs, float* mu_q, float* sd_q, float* mu_p, float* sd_p, float* reserve, void* stream);
/*   d_z [Tp,B,Z]: gradient wrt z_t from outside the chain (the decoder).  KL folded in as for blvm_vrnn_seq_bwd.
 *   Outputs d_d, d_a [Tp,B,R], d_z0 [B,Z] (each may be NULL); grads accumulated. */
int blvm_srnn_latent_bwd(const BlvmSrnnWeights* w, const float* d, const float* a, const float* eps, const float* zs,
                         const float* mu_q, const float* sd_q, const float* mu_p, const float* sd_p,
                         const float* reserve, const float* d_z, const int32_t* x_sl, const float* c_raw,
                         const float* c_fn, int stride, float fn_floor, int Tp, int B, int H, int Z, int R,
                         int residual_posterior, float sd_eps, float slope, float* d_d, float* d_a, float* d_z0,
                         const BlvmSrnnGrads* grads, float* workspace, void* stream);

/* K3c  Ancestral sampling from SRNNAudio: every step of every utterance in ONE persistent launch (B <= 128).  Replaces the loop of
 *   `SRNN.generate` (`blvm/models/srnn.py:304-403`): enc = encoder(x_t) -> d_t = GRU(enc, d_{t-1}) -> prior(cat[d_t, z_{t-1}]) ->
 *   z_t = mu + sd eps -> decoder(cat[z_t, d_t]) -> DMoL head per sample -> draw -> x_{t+1}; 13 links per step dealt over all CUs
 *   (csrc/pchain.hip).  Weights in their PyTorch layouts; encoder / decoder are 3 x (Linear + LeakyReLU(slope)) as in
 *   `srnn.py:456-474`, x_dim = H.  x0 [B,S]; d0 [B,R], z0 [B,Z] or NULL (zeros); eps [T,B,Z]; u [T,B,S,num_mix], v [T,B,S] as in
 *   blvm_mix_sample (both NULL: the mode).  Outputs: x_out [B,T,S]; d_out [B,R] = d_T and z_out [T,B,Z] (each may be NULL).
 *   scratch: blvm_srnn_generate_scratch_floats(...) floats (weight copies + one slab per step of every activation). */
typedef struct BlvmSrnnDecodeWeights {
  const float *enc_w[3], *enc_b[3];                    /* [H,S], [H,H], [H,H] */
  const float *gru_wih, *gru_whh, *gru_bih, *gru_bhh;  /* d_forward_recurrent: [3R,H], [3R,R], [3R], [3R] */
  const BlvmSrnnWeights* chain;                        /* prior_* are read */
  const float *dec_w[3], *dec_b[3];                    /* [H,Z+R] (input order cat[z, d]), [H,H], [S*3*num_mix,H] */
  const float *lik_w, *lik_b;                          /* [3*num_mix, 3*num_mix], [3*num_mix] */
} BlvmSrnnDecodeWeights;
size_t blvm_srnn_generate_scratch_floats(int T, int B, int S, int H, int Z, int R);
int blvm_srnn_generate(const BlvmSrnnDecodeWeights* w, const float* x0, const float* d0, const float* z0, const float* eps,
                       const float* u, const float* v, int T, int B, int S, int H, int Z, int R, int num_mix, float sd_eps,
                       float slope, float log_eps, float* x_out, float* d_out, float* z_out, float* scratch, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K10  WaveNet: dilated causal convolution (kernel size 2) and the gated residual block.  Replaces
 *      `CausalConv1d` (`blvm/models/wavenet/wavenet_modules.py:14-50`) and `Conv1dResidualGLU` (`:53-117`).
 *      Activations are TIME-MAJOR channel-last [L, B, C]; weights keep the Conv1d layout [C_out, C_in, k].
 *      out[t] = W[:,:,0] x[t] + W[:,:,1] x[t+dilation] + bias for t in [0, L_in - dilation)  (no padding: a block
 *      consumes `dilation` frames on the left, as the reference's stack does).
 * ------------------------------------------------------------------------------------------------------------- */
int blvm_scale_act_f32(const float* x, float scale, float slope, float* y, size_t n, void* stream); /* y = act(scale x) */
size_t blvm_conv1d_k2_workspace_floats(int Cin, int Cout);
int blvm_conv1d_k2_fwd(const float* x, const float* W, const float* bias, int L_in, int B, int Cin, int Cout,
                       int dilation, float* out, float* workspace, void* stream);
/*   d_x [L_in,B,Cin] (=), dW [Cout,Cin,2] (+=), db [Cout] (+=); each may be NULL. */
int blvm_conv1d_k2_bwd(const float* x, const float* W, const float* d_out, int L_in, int B, int Cin, int Cout,
                       int dilation, float* d_x, float* dW, float* db, float* workspace, void* stream);

/*   x [L_in,B,C]; conv_w [2C,C,2], conv_b [2C]; rs_w [C+S,C] (the 1x1 conv: first C rows residual, last S rows skip),
 *   rs_b [C+S]; o [L_in-dilation,B,C] = (res + x[dilation:]) * inv_std (may be NULL for the last block);
 *   skip [T_skip,B,S] += the last T_skip frames of the skip branch; S == 0 (skip NULL, rs_w = the first C rows): a block
 *   whose skip output is unused (STCN reads only every n-th block's skip, `stcn.py:301`).  reserve keeps pre-activations. */
size_t blvm_wavenet_block_reserve_floats(int L_in, int B, int C, int dilation);
size_t blvm_wavenet_block_workspace_floats(int L_in, int B, int C, int S, int dilation);
int blvm_wavenet_block_fwd(const float* x, const float* conv_w, const float* conv_b, const float* rs_w,
                           const float* rs_b, int L_in, int B, int C, int S, int dilation, int T_skip, float inv_std,
                           float* o, float* skip, float* reserve, float* workspace, void* stream);
/*   d_o may be NULL (last block); d_x [L_in,B,C] (=); weight grads (+=), each may be NULL. */
int blvm_wavenet_block_bwd(const float* x, const float* conv_w, const float* rs_w, const float* reserve,
                           const float* d_o, const float* d_skip, int L_in, int B, int C, int S, int dilation,
                           int T_skip, float inv_std, float* d_x, float* dconv_w, float* dconv_b, float* drs_w,
                           float* drs_b, float* workspace, void* stream);

/*   The whole residual stack in one call (`ResidualStack.forward`, `blvm/models/wavenet/wavenet_modules.py:178-215`: the loop over
 *   `Conv1dResidualGLU` blocks and the sum of their skip outputs): the host loop over blvm_wavenet_block_fwd / _bwd inside the library.
 *   params: HOST array of 4 n_blocks device pointers (conv_w, conv_b, rs_w, rs_b per block); dilations, groups: HOST arrays
 *   [n_blocks]; groups[i] = index into skips / d_skips of the tensor block i's skip branch is accumulated into, -1 = unused
 *   (that block runs with S = 0).  acts: the residual outputs of blocks 0 .. n-2 back to back, reserve: the blocks' reserves back
 *   to back (sizes from blvm_wavenet_stack_floats).  Backward: d_x [L,B,C] (=), d_scratch [L,B,C] (overwritten), grads: HOST array
 *   of 4 n_blocks device pointers (+=, each may be NULL). */
int blvm_wavenet_stack_floats(int L, int B, int C, const int* dilations, int n_blocks, size_t* acts_floats, size_t* reserve_floats);
int blvm_wavenet_stack_fwd(const float* x, const float* const* params, const int* dilations, const int* groups, int n_blocks, int L,
                           int B, int C, int S, int T_skip, float inv_std, float* acts, float* const* skips, float* reserve,
                           float* workspace, void* stream);
int blvm_wavenet_stack_bwd(const float* x, const float* const* params, const int* dilations, const int* groups, int n_blocks, int L,
                           int B, int C, int S, int T_skip, float inv_std, const float* acts, const float* reserve,
                           const float* const* d_skips, float* d_x, float* d_scratch, float* const* grads, float* workspace,
                           void* stream);

/* K10c  Autoregressive WaveNet sampling: all frames of all utterances in ONE launch, from an all-zero start.  Replaces the
 *   per-frame loop of `WaveNet.generate` (`blvm/models/wavenet/wavenet.py:254-293`) with the cached formulation its TODO
 *   names: block i keeps a ring buffer of its input over the last dilation_i frames.  in_channels = n_stack_frames = 1.
 *   packed: one weight image of blvm_wavenet_decode_pack_floats(C,S,O,n_blocks) floats, in this order (row-major,
 *     PyTorch parameter layouts): causal conv w [C,1,2], b [C]; in_transform w [C,C], b [C]; per block conv w [2C,C,2],
 *     b [2C], rs w [C+S,C], b [C+S]; out_transform Linear w [O,S], b [O]; head Linear w zero-padded to [32,O], b to [32].
 *   dilations: HOST array [n_blocks], n_blocks <= 64.  C, S, O multiples of 16; 3*num_mix <= 32.
 *   Per frame: skip sum * skip_scale -> ReLU -> Linear -> ReLU -> head Linear -> (logits, locs, log_scales) -> Gumbel-max
 *   component pick with u [n_frames,B,num_mix], logistic draw with v [n_frames,B] clamped to [-1,1] (as blvm_mix_sample
 *   kind 0; u = v = NULL: the mode) -> x_out [B,n_frames], fed back as the next input.
 *   scratch: blvm_wavenet_decode_scratch_floats(...) floats (ring buffers + operand-layout copies of the block weights;
 *   contents need no initialisation). */
size_t blvm_wavenet_decode_pack_floats(int C, int S, int O, int n_blocks);
size_t blvm_wavenet_decode_scratch_floats(const int* dilations, int n_blocks, int B, int C, int S);
int blvm_wavenet_decode(const float* packed, const int* dilations, int n_blocks, int B, int C, int S, int O, int num_mix,
                        int n_frames, float inv_std, float skip_scale, float log_eps, const float* u, const float* v,
                        float* scratch, float* x_out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K5  RSSM cell of the Clockwork-VAE over a sequence (forward + BPTT).  Replaces the per-level time loop
 *     `blvm/models/clockwork_vae/clockwork_vae.py:272-281` over `RSSMCell.forward` (`blvm/modules/rssm.py:79-104`).
 *   enc [T,B,E] encodings of this level; ctx [T,B,C] context from the level above (C may be 0, ctx NULL);
 *   z0 [B,Z], h0 [B,H] or NULL (zeros); eps [T,B,Z];
 *   mode: 0 plain, 1 residual posterior (mu_q += mu_p), 2 precision-weighted posterior (`variational.py:125-138`).
 *   zs [T+1,B,Z], hs [T+1,B,H]: row 0 = initial state, row t+1 = state after step t (the context for the level below
 *   is cat[zs[1:], hs[1:]]).  mu_q/sd_q are the COMBINED posterior parameters the KL is taken against.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct BlvmRssmWeights {
  const float *gin_w, *gin_b;           /* [H, Z+C] (input order cat[z, context]), [H] */
  const float *gru_wih, *gru_whh;       /* [3H,H] each, rows [r|z|n] */
  const float *gru_bih, *gru_bhh;       /* [3H] */
  const float *prior_w[3], *prior_b[3]; /* [H,H] x3 */
  const float *prior_hw, *prior_hb;     /* [2Z,H] */
  const float *post_w[3], *post_b[3];   /* [H,H+E] (input order cat[h, enc]), [H,H], [H,H] */
  const float *post_hw, *post_hb;       /* [2Z,H] */
} BlvmRssmWeights;

typedef struct BlvmRssmGrads { /* same shapes; ACCUMULATED into (caller zeroes) */
  float *gin_w, *gin_b, *gru_wih, *gru_whh, *gru_bih, *gru_bhh;
  float *prior_w[3], *prior_b[3], *prior_hw, *prior_hb;
  float *post_w[3], *post_b[3], *post_hw, *post_hb;
} BlvmRssmGrads;

size_t blvm_rssm_reserve_floats(int T, int B, int H, int Z);
size_t blvm_rssm_bwd_workspace_floats(int T, int B, int H, int Z);
int blvm_rssm_seq_fwd(const BlvmRssmWeights* w, const float* enc, const float* ctx, const float* z0, const float* h0,
                      const float* eps, int T, int B, int H, int Z, int C, int E, int mode, float sd_eps, float* zs,
                      float* hs, float* mu_q, float* sd_q, float* mu_p, float* sd_p, float* reserve, void* stream);
/*   d_zs [T+1,B,Z], d_hs [T+1,B,H]: gradients wrt zs / hs from outside the chain (rows 1.. = the states, row 0 = the
 *   initial state).  KL folded in as for blvm_vrnn_seq_bwd (stride = level stride in audio frames).
 *   Outputs (each may be NULL): d_enc [T,B,E], d_ctx [T,B,C], d_z0 [B,Z] (complete), d_h0 [B,H] (WITHOUT d_hs[0]). */
int blvm_rssm_seq_bwd(const BlvmRssmWeights* w, const float* enc, const float* ctx, const float* eps, const float* zs,
                      const float* hs, const float* mu_q, const float* sd_q, const float* mu_p, const float* sd_p,
                      const float* reserve, const float* d_zs, const float* d_hs, const int32_t* x_sl,
                      const float* c_raw, const float* c_fn, int stride, float fn_floor, int T, int B, int H, int Z,
                      int C, int E, int mode, float sd_eps, float* d_enc, float* d_ctx, float* d_z0, float* d_h0,
                      const BlvmRssmGrads* grads, float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * K11  Clockwork-VAE convolutional coders: the streaming pieces of `BlockSeparable`
 * (`blvm/models/clockwork_vae/convolutional_coders.py:29-66`), `ConvDepthwiseSeparable1d` /
 * `ConvTransposeDepthwiseSeparable1d` (`blvm/modules/convolutions.py:6-104`) and `TemporalResidual`
 * (`convolutional_coders.py:15-26`).  Layout: time-major channel-last [L,B,C] seen as a matrix [L, N=B*C]
 * (C a multiple of 4).  The block's 1x1 convolutions are blvm_gemm_f32 calls on [L*B, C].
 *
 * chan_norm: nn.GroupNorm(num_groups=C, C) == per-(sample,channel) normalisation over time, biased variance, eps;
 *   y = (x-mean)*rstd*gamma_c + beta_c;  mr [2,N] receives [mean | rstd] (kept for backward);
 *   workspace: blvm_chan_norm_workspace_doubles(N) float64.
 *   blvm_chan_norm_stats: statistics only — mr and the column affine scale_shift [2,N] = [rstd*gamma | beta - mean*rstd*gamma]
 *   for consumers that normalise on load (blvm_dwconv_*'s in_scale / in_shift), so the normalised tensor never hits HBM.
 *   bwd: dx (relu_mask=1 multiplies by (x>0): a ReLU feeding the norm), dgamma/dbeta ACCUMULATED (may be NULL),
 *   dx_chan_sum [C] (may be NULL) += sum over rows and samples of dx (the bias gradient of the layer that produced x).
 * dwconv: depthwise Conv1d (transposed=0: L_out=(L_in-k_eff)/stride+1) or ConvTranspose1d (transposed=1:
 *   L_out=(L_in-1)*stride+k_eff), k<=8, no padding (callers pad), weights [C,k] (= torch [C,1,k]), bias [C] or NULL,
 *   relu=1 fuses a ReLU on the output (bwd then needs y).  in_scale/in_shift ([N] each, or both NULL): the convolution
 *   runs on x*in_scale + in_shift (column-wise).  bwd: dx = gradient wrt that transformed input (may be NULL),
 *   dw/dbias ACCUMULATED (may be NULL).
 * resample_add: out[t] = y[t] + x[nearest(t)], nearest(t)=min(floor(t*float(L_in)/L_out), L_in-1) (F.interpolate
 *   mode='nearest');  bwd ACCUMULATES dout into dx [L_in,N] (dy is dout itself).
 * ------------------------------------------------------------------------------------------------------------- */
size_t blvm_chan_norm_workspace_doubles(int N);
int blvm_chan_norm_stats(const float* x, int L, int N, int C, const float* gamma, const float* beta, float eps, float* mr,
                         float* scale_shift, double* workspace, void* stream);
int blvm_chan_norm_fwd(const float* x, int L, int N, int C, const float* gamma, const float* beta, float eps, float* y,
                       float* mr, double* workspace, void* stream);
int blvm_chan_norm_bwd(const float* x, const float* dy, const float* mr, const float* gamma, int L, int N, int C,
                       int relu_mask, float* dx, float* dgamma, float* dbeta, float* dx_chan_sum, double* workspace,
                       void* stream);
int blvm_dwconv_out_length(int L_in, int k, int stride, int dilation, int transposed);
int blvm_dwconv_fwd(const float* x, const float* in_scale, const float* in_shift, const float* w, const float* bias,
                    int L_in, int N, int C, int k, int stride, int dilation, int transposed, int relu, float* y,
                    void* stream);
int blvm_dwconv_bwd(const float* x, const float* in_scale, const float* in_shift, const float* w, const float* y,
                    const float* dy, int L_in, int N, int C, int k, int stride, int dilation, int transposed, int relu,
                    float* dx, float* dw, float* dbias, void* stream);
int blvm_resample_add_fwd(const float* y, const float* x, int L_out, int L_in, int N, float* out, void* stream);
int blvm_resample_add_bwd(const float* dout, int L_out, int L_in, int N, float* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BLVM_HIP_H */
