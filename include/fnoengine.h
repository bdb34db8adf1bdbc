/* fnoengine C ABI - MI355X (gfx950) FNO spectral-convolution engine.
 *
 * Drop-in boundary for the hot path of neuraloperator/pde-policylearning
 * (SURVEY.md section 8b).  The reference is pure Python/PyTorch, so "what its FFI
 * would bind" is the set of torch-op sequences below; each entry point names the
 * reference code it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs; no torch / C++ types.
 *   - every pointer is DEVICE memory owned by the caller (activations NCHW /
 *     NCDHW contiguous fp32, complex tensors interleaved (re, im) fp32) unless
 *     stated otherwise.  The library owns only immutable twiddle tables inside
 *     a plan (freed by *_plan_destroy).
 *   - all work is enqueued on the caller's hipStream_t (passed as void*), no
 *     implicit synchronisation, no allocation inside forward / backward
 *     (graph-capturable).
 *   - return 0 on success, negative FNO_E* on failure; text via fno_last_error().
 */
#ifndef FNOENGINE_H
#define FNOENGINE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FNO_MAX_LAYERS 16
#define FNO_VERSION 100

enum { FNO_OK = 0, FNO_EINVAL = -1, FNO_EUNSUPPORTED = -2, FNO_EHIP = -3, FNO_ENOMEM = -4 };
/* torch.fft `norm=` strings */
enum { FNO_NORM_BACKWARD = 0, FNO_NORM_FORWARD = 1, FNO_NORM_ORTHO = 2 };

int fno_version(void);
const char* fno_last_error(void);
/* GEMM arithmetic of the fused model kernels.
 * 1 (default) = split precision, fp32 in / fp32 accumulate / fp32 out: every operand of a channel GEMM is scaled by a power of
 *     two taken from a published bound of its magnitude and split into TWO fp16 terms x = h + l (22 significand bits); the
 *     products h h', h l', l h' run on the fp16 matrix pipe (v_mfma_f32_32x32x16_f16: THREE products per 16-deep k block),
 *     hh and the cross terms in separate fp32 accumulators.  Where no bound is known before the operand is split (the
 *     spectral K-extension's table and spectral rows, small-K extensions, kernels of models with fewer than 1024 tiles) the
 *     operand is split into THREE bf16 terms instead and six products are kept (v_mfma_f32_32x32x16_bf16).  Error of either
 *     form against float64 = the fp32 MFMA's own (1.5e-7 relative at K = 64; DESIGN.md sections 4, 4d).
 * 0 = every GEMM on the exact fp32 matrix instruction (v_mfma_f32_32x32x2_f32).
 * Environment FNO_GEMM_F32=1 selects 0 at load time. */
void fno_set_gemm_mode(int split_precision);
int fno_get_gemm_mode(void);
/* The mode contraction 'bixy,ioxy->boxy' (spectral_convolution.py:15-36, rno.py:51-58, basics.py:14-24) and its two
 * adjoints: 1 (default) = one real GEMM per kept mode on the fp32 matrix cores (32 / 64 channels; other channel counts
 * always take the VALU kernels); 0 = VALU kernels everywhere.  Environment FNO_MODE_GEMM_VALU=1 selects 0 at load time. */
void fno_set_mode_gemm(int mfma);
int fno_get_mode_gemm(void);

/* The spectral middle of a fused 2-D block (leading-axis DFT -> mode contraction -> leading-axis inverse DFT,
 * spectral_convolution.py:324-345): 1 (default) = one launch (k_spec_mid; 32 / 64 channels, 4 / 8 / 12 / 16 kept leading
 * modes), 0 = the three-launch sequence.  Same results to fp32 rounding.  Environment FNO_NO_FUSED_MID=1 selects 0 at
 * load time. */
void fno_set_fused_mid(int on);
int fno_get_fused_mid(void);

/* ------------------------------------------------------------------------
 * Standalone spectral convolution  y = irfftn(pad(W_c . rfftn(x)[corner_c])) (+ bias)
 * Covers the reference's three dialects:
 *   A  neuralop/models/spectral_convolution.py:303-347 (FactorizedSpectralConv, dense;
 *      modes[d] = n_modes[d] // 2, norm FORWARD via FNO default tfno.py:129, bias)
 *   B  neuralop/models/rno.py:60-77          (SpectralConv2d, modes as given, ORTHO)
 *   C  libs/models/pino_models/basics.py:79-96, 114-143 (SpectralConv2d/3d, BACKWARD;
 *      3-D: weight_last_extent = modes3, modes[2] = min(Nz/2+1, modes3))
 * Corner weights are the reference's own parameter tensors, complex64 viewed as
 * fp32 pairs, shape (Cin, Cout, modes[0], [modes[1],] weight_last_extent), in the
 * canonical corner order (lo), (hi) in 2-D and (lo,lo), (lo,hi), (hi,lo), (hi,hi)
 * over the two leading dims in 3-D.
 * ---------------------------------------------------------------------- */
typedef struct FnoSpecDesc {
  int ndim;                /* 2 or 3 */
  int Cin, Cout;
  int dims[3];             /* spatial extents */
  int modes[3];            /* kept extent per corner along each dim */
  int weight_last_extent;  /* last-dim extent of the stored weights (>= modes[ndim-1]) */
  int norm;                /* FNO_NORM_* */
  int input_gelu;          /* 1: the input is a PRE-activation tensor u and the convolution acts on gelu(u) (applied while
                              the rows are staged; <= 64 channels, rows <= 320 floats, <= 32 kept last-dim bins).  backward
                              then returns dL/d gelu(u); the caller (fno_pointwise_backward) applies gelu'(u). */
  int weight_planes;       /* 1: the corner weights (and their gradients) are stored PLANE-MAJOR - the memory order is
                              (weight_last_extent, Cin, Cout, modes[0], [modes[1]]), i.e. the reference's tensor permuted so
                              that its last dim is outermost.  The live last-dim slices [0, modes[ndim-1]) of a dialect-C
                              weight are then one contiguous prefix (what the layout kernels, the optimizer and the gradient
                              exchange touch: PINObserverFullField at T = 1 uses 1 plane of 12).  fno_spec_backward then
                              writes the live planes of dw_corners only: the caller keeps the others zero. */
} FnoSpecDesc;

typedef struct FnoSpecPlan FnoSpecPlan;
int fno_spec_plan_create(const FnoSpecDesc* desc, FnoSpecPlan** out);
void fno_spec_plan_destroy(FnoSpecPlan* plan);
size_t fno_spec_workspace_bytes(const FnoSpecPlan* plan, int batch);
size_t fno_spec_xhat_bytes(const FnoSpecPlan* plan, int batch);
/* y = specconv(x) + bias.  xhat_save (fno_spec_xhat_bytes) receives what backward needs besides dy: the truncated
 * spectrum of x (for dW) and the mode-major transposed weights (for dx, so that backward does not re-pack them);
 * may be NULL.  fno_spec_backward with xhat = that buffer may pass w_corners = NULL. */
int fno_spec_forward(const FnoSpecPlan* plan, int batch, const float* x, const float* const* w_corners,
                     const float* bias /*nullable (Cout)*/, float* y, float* xhat_save, void* ws, size_t ws_bytes,
                     void* stream);
/* autograd of the above: any of dx / dw_corners / dbias may be NULL. */
int fno_spec_backward(const FnoSpecPlan* plan, int batch, const float* dy, const float* xhat_save,
                      const float* const* w_corners, float* dx, float* const* dw_corners, float* dbias, void* ws,
                      size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Whole-model fused path: neuralop.models.FNO on its default configuration
 * (tfno.py:195-211: Lifting :11-20 -> n_layers x FNOBlocks.forward fno_block.py:123-170
 * with fno_skip='linear', GELU gate `index < n_layers - index` :149 -> Projection :23-38).
 * One fused kernel per block (skip 1x1 conv + last-dim inverse DFT + bias + gated GELU +
 * last-dim forward DFT of the next block), activations stored pre-activation.
 * Grids: hidden width 32 or 64; rows (last dim) of 32 / 64 / 128 / 256 tile the kernels' 128- or 256-pixel tiles and take the
 * row transforms as kernel epilogues; any other last dim in 32..320 on planes that are a multiple of 128 pixels (96 x 96,
 * 160 x 160, the PINO observers' padded time axis 73, ...) runs in "loose rows" mode: 128-pixel tiles of the flattened plane,
 * spectral rows gathered per tile, last-dim forward transforms as separate passes (split-precision GEMM mode only).
 * fno_model_plan_create returns FNO_EUNSUPPORTED for everything else; callers then compose fno_spec_* / fno_pointwise_*.
 * ---------------------------------------------------------------------- */
typedef struct FnoModelDesc {
  int ndim;            /* 2 or 3 */
  int Cin, C, Cout;    /* lifting in, hidden width, projection out */
  int hidden_proj;     /* projection_channels (256) */
  int n_layers;
  int dims[3];
  int modes[3];        /* kept per corner per dim = n_modes[d] // 2 (spectral_convolution.py:202-203) */
  int norm;            /* FNO_NORM_* (FNO default: FORWARD) */
  unsigned gelu_mask;  /* bit l set <=> GELU after block l (fno_block.py:149) */
  int weight_planes;   /* 1: spec_w tensors (and their gradients) are stored plane-major, see FnoSpecDesc */
} FnoModelDesc;

typedef struct FnoModelParams {       /* all fp32 device pointers, reference parameter layouts */
  const float* lift_w;                /* lifting.fc.weight (C, Cin, 1..)  */
  const float* lift_b;                /* lifting.fc.bias (C)              */
  const float* skip_w[FNO_MAX_LAYERS];      /* fno_blocks.fno_skips.l.weight (C, C, 1..) */
  const float* spec_w[FNO_MAX_LAYERS][4];   /* fno_blocks.convs.weight[2^(d-1) l + corner] (C, C, m.., 2) */
  const float* spec_bias;             /* fno_blocks.convs.bias (L, C, 1..) or NULL */
  const float* proj_w1;               /* projection.fc1.weight (hidden_proj, C, 1..) */
  const float* proj_b1;
  const float* proj_w2;               /* projection.fc2.weight (Cout, hidden_proj, 1..) */
  const float* proj_b2;
} FnoModelParams;

typedef struct FnoModelGrads {        /* same shapes as the parameters; written, not accumulated */
  float* lift_w;
  float* lift_b;
  float* skip_w[FNO_MAX_LAYERS];
  float* spec_w[FNO_MAX_LAYERS][4];
  float* spec_bias;
  float* proj_w1;
  float* proj_b1;
  float* proj_w2;
  float* proj_b2;
} FnoModelGrads;

typedef struct FnoModelPlan FnoModelPlan;
int fno_model_plan_create(const FnoModelDesc* desc, FnoModelPlan** out);
void fno_model_plan_destroy(FnoModelPlan* plan);
size_t fno_model_workspace_bytes(const FnoModelPlan* plan, int batch);
/* bytes of the forward->backward stash: (n_layers+1) pre-activation tensors +
 * n_layers truncated spectra */
size_t fno_model_saved_bytes(const FnoModelPlan* plan, int batch);
int fno_model_forward(const FnoModelPlan* plan, int batch, const FnoModelParams* p, const float* x, float* y,
                      void* saved /*nullable: inference*/, void* ws, size_t ws_bytes, void* stream);
int fno_model_backward(const FnoModelPlan* plan, int batch, const FnoModelParams* p, const float* x,
                       const float* dy, const void* saved, const FnoModelGrads* g, void* ws, size_t ws_bytes,
                       void* stream);
/* Block stacks.  FnoModelDesc.Cin == 0 drops the lifting (x is the (B, C, ...) input of block 0) and
 * Cout == 0 drops the projection (y = u_L, (B, C, ...); the last block must then carry no GELU).  This is
 * the fused form of the reference's other "Fourier layers":
 *   neuralop/models/rno.py:215-228  FourierLayer2d = SpectralConv2d(ortho) + Conv1d(k=1)   (n_layers = 1)
 *   libs/models/pino_models/pinobserver.py:221-226  sp_convs[i](x) + ws[i](x), GELU except last
 * with skip_w[l] = the Conv1d weight and spec_bias = the Conv1d biases (L, C).  For such stacks the
 * input gradient dx (B, C, ...) is produced as well (NULL to skip). */
int fno_model_backward_dx(const FnoModelPlan* plan, int batch, const FnoModelParams* p, const float* x,
                          const float* dy, const void* saved, const FnoModelGrads* g, float* dx, void* ws,
                          size_t ws_bytes, void* stream);
/* One-layer block stacks with a tail: the layers of the RNO regressor, neuralop/models/rno.py:92-106
 * SpectralConvWithFC.forward = act(spec_conv(dropout(x)) + Linear(x)) with act = ReLU (rno.py:214-215), channels-first.
 *   forward   y = max(u, 0) if relu_out else u,   u = specconv(drop(x)) + skip_w x + bias
 *   backward  g = dy * (y > 0) (the mask is read off the forward's output `y`), dx = skip_w^T g + drop-scale * (spectral
 *             adjoint of g); parameter gradients as fno_model_backward_dx
 * drop(x)[e] = x[e] * s[e], s[e] = 0 with probability drop_p and 1 / (1 - drop_p) otherwise, decided by a hash of the
 * element index and the two 32-bit words at `drop_seed` (device memory; the caller draws them per call and passes the same
 * pointer to the backward, which regenerates s instead of reading a mask; fno_dropout_scale writes s out for tests).
 * drop_p = 0 disables the dropout (evaluation mode).  Plans: Cin = Cout = 0, n_layers = 1, 32 / 64 channels, rows of
 * 32 / 64 / 128 floats, split-precision GEMM mode; anything else returns FNO_EUNSUPPORTED. */
typedef struct FnoBlockTail {
  int relu_out;
  float drop_p;
  const unsigned* drop_seed;   /* device pointer to 2 x uint32, or NULL when drop_p == 0 */
  const float* y;              /* backward only: the forward's output (B, C, ...) */
} FnoBlockTail;
int fno_model_forward_tail(const FnoModelPlan* plan, int batch, const FnoModelParams* p, const float* x, float* y,
                           void* saved, void* ws, size_t ws_bytes, void* stream, const FnoBlockTail* tail);
int fno_model_backward_tail(const FnoModelPlan* plan, int batch, const FnoModelParams* p, const float* x,
                            const float* dy, const void* saved, const FnoModelGrads* g, float* dx, void* ws,
                            size_t ws_bytes, void* stream, const FnoBlockTail* tail);
int fno_dropout_scale(size_t n, float drop_p, const unsigned* seed /*device, 2 words*/, float* out /*device, n*/, void* stream);

/* The same pass in parts: layers l_hi .. l_lo (descending; l_hi = n_layers-1 includes the projection, l_lo = 0 the
 * lifting).  Calls over a partition of the layers with the SAME workspace reproduce the full pass bit for bit, and each
 * call finishes the gradients of its own layers - a data-parallel caller starts the all-reduce of the late layers'
 * gradients while the early layers are still being differentiated (trainer.FlatGradBucket.for_fno). */
int fno_model_backward_part(const FnoModelPlan* plan, int batch, const FnoModelParams* p, const float* x,
                            const float* dy, const void* saved, const FnoModelGrads* g, float* dx, void* ws,
                            size_t ws_bytes, void* stream, int l_hi, int l_lo);

/* Fan-out of Fourier layers over ONE input: y_j = SpecConv_j(x) + W_j x + b_j, j < n_out.  The GRU-style cell of the
 * recurrent neural operator evaluates eight Fourier layers on three distinct inputs (neuralop/models/rno.py:254-260:
 * f1, f3, f5, f7 on x; f2, f4, f8 on h; f6 on r * h); the reference transforms each input once per layer.  Here the
 * forward transforms of the shared input run once for all members, and backward produces dx = sum_j dL/dx|_j by chaining
 * the members through the block kernel's gradient-addend input (no accumulation pass, no n_out separate dx tensors).
 * `plan` is a block-stack plan (Cin = Cout = 0, gelu_mask = 0) created with n_layers >= n_out; members occupy layer slots
 * 0..n_out-1 of FnoModelParams / FnoModelGrads (skip_w, spec_w only), biases are separate (C) arrays (entries / the
 * array itself nullable).  The members' mode contractions and leading-axis passes run as ONE launch each (member-major
 * buffers, member index on the contraction grid).  Workspace: fno_fanout_workspace_bytes; stash: fno_fanout_saved_bytes. */
size_t fno_fanout_saved_bytes(const FnoModelPlan* plan, int batch, int n_out);
size_t fno_fanout_workspace_bytes(const FnoModelPlan* plan, int batch, int n_out);
int fno_fanout_forward(const FnoModelPlan* plan, int batch, int n_out, const FnoModelParams* p, const float* const* bias,
                       const float* x, float* const* y, void* saved, void* ws, size_t ws_bytes, void* stream);
int fno_fanout_backward(const FnoModelPlan* plan, int batch, int n_out, const FnoModelParams* p, const float* x,
                        const float* const* dy, const void* saved, const FnoModelGrads* g, float* const* dbias, float* dx,
                        void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Training-step tail (run_pde_observers.py:185-193), SURVEY.md section 8(f) rank 2.
 * Loss: pd = pred*(std+eps)+mean, td = target*(std+eps)+mean  (NormalizerGivenMeanStd.cuda_decode,
 * libs/utilities3.py:115-129; mean/std NULL = identity; stat_len = 1 or n, broadcast over the batch),
 * loss = sum_b ||pd_b - td_b||_2 / ||td_b||_2, divided by batch when size_average
 * (LpLoss.rel, libs/utilities3.py:323-334).  forward leaves per-sample coefficients in `ws`
 * (fno_lploss_workspace_bytes) for backward, which writes dloss/dpred scaled by the device scalar
 * *grad_loss (NULL = 1).  No host synchronisation (the reference's loss.item() is the caller's choice).
 * ---------------------------------------------------------------------- */
size_t fno_lploss_workspace_bytes(int batch);
int fno_lploss_rel_forward(int batch, size_t n_per_sample, const float* pred, const float* target, const float* mean,
                           const float* std, int stat_len, float eps, int size_average, float* loss /*device scalar*/,
                           void* ws, size_t ws_bytes, void* stream);
int fno_lploss_rel_backward(int batch, size_t n_per_sample, const float* pred, const float* target, const float* std,
                            int stat_len, float eps, const float* grad_loss, float* dpred, const void* ws,
                            size_t ws_bytes, void* stream);
/* torch.optim.Adam (run_pde_observers.py:134: lr, weight_decay as L2 added to the gradient; no amsgrad)
 * on ONE flat fp32 bucket of n elements: param / grad / exp_avg / exp_avg_sq, 16-byte aligned.
 * `step` is the 1-based step count.  The hyperparameters are DOUBLES, as torch.optim.Adam holds them: step size, bias
 * corrections and the (1 - beta) weights are formed in double and rounded once, as torch's scalar arguments are
 * (1.0f - 0.999f is 1.3e-5 away from the 0.001 torch hands to addcmul_). */
int fno_adam_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int step, void* stream);
/* Same update with the step count on the DEVICE: *step_counter is incremented and the bias corrections are
 * derived from it by a one-thread kernel, so a captured hipGraph of the whole training step can be
 * replayed (no host-side scalar changes between replays).  scratch2: two device floats. */
int fno_adam_step_dev(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                      double beta2, double eps, double weight_decay, int* step_counter, float* scratch2, void* stream);
/* The same update for a bucket that holds dialect-C spectral weights of which only the last-dim slice [..., :k] ever sees
 * data (libs/models/pino_models/basics.py:119-139 cuts the spectrum at min(Nz/2+1, modes3): PINObserverFullField at T = 1
 * uses 1/12 of its 906 MB).  The gradient of the rest is exactly zero, so what torch.optim.Adam (run_pde_observers.py:134)
 * does to such an element is a recurrence on (p, m, v) alone - g = weight_decay * p.  It is skipped per step and REPLAYED in
 * one pass when the slice is needed; results are bit-identical to stepping it every time.
 *   A sliced block is rows x row_len floats, the first live_len floats of each row live (both even: complex pairs);
 *   a plane-major weight (FnoSpecDesc.weight_planes) is ONE row whose live planes are its head.
 *   fno_adam_step_range: fno_adam_step on a sub-range; `dyn` (device, 2 floats: fno_adam_prep_dev) replaces `step`.
 *   fno_adam_prep_dev:   ++*step_counter and the step's two scalars into scratch2 (graph-replayable, once per step).
 *   fno_adam_step_live:  the live part of a block; param / grad in the full layout, moments compact (rows x live_len).
 *   fno_adam_scalars:    host: {lr / (1 - beta1^step), sqrt(1 - beta2^step)} exactly as fno_adam_step derives them.
 *   fno_adam_replay_prep: the same two scalars for steps step_from .. step_from + nsteps - 1 on the device
 *                        (scal: 2 * nsteps floats), as fno_adam_step_dev derives them.
 *   fno_adam_replay_dead: takes the dead part of a block through the nsteps steps described by `scal`; dead moments
 *                        compact (rows x (row_len - live_len)), read unless moments_zero, always written. */
void fno_adam_scalars(double lr, double beta1, double beta2, int step, float* out2);
int fno_adam_prep_dev(int* step_counter, float* scratch2, double lr, double beta1, double beta2, void* stream);
int fno_adam_step_range(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                        double beta2, double eps, double weight_decay, int step, const float* dyn, void* stream);
int fno_adam_step_live(size_t rows, int row_len, int live_len, float* param, const float* grad, float* exp_avg_live,
                       float* exp_avg_sq_live, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                       const float* dyn, void* stream);
int fno_adam_replay_prep(float* scal, int step_from, int nsteps, double lr, double beta1, double beta2, void* stream);
int fno_adam_replay_dead(size_t rows, int row_len, int live_len, float* param, float* dead_exp_avg, float* dead_exp_avg_sq,
                         int moments_zero, const float* scal, int nsteps, double beta1, double beta2, double eps,
                         double weight_decay, void* stream);

/* ------------------------------------------------------------------------
 * PINO residual loss, SURVEY.md section 8(f) rank 1: FDM_NS_vorticity + Channelflow_PINO_loss
 * (libs/envs/diff_control_env.py:5-60 == libs/pino_utils/losses.py:68-104, 246-262; train_pino.py:98-101).
 *   u (B, n, n, nt) model output, u0 (B, n, n) initial vorticity, forcing (n, n), visc (B) = 1 / Re,
 *   Du = w_t + u . grad(w) - visc lap(w) with spectral derivatives over (x, y) and a central difference in t
 *   on the interior time levels; loss_f = mean_b ||Du_b - forcing|| / ||forcing||, loss_ic = mean_b
 *   ||u_b(t = 0) - u0_b|| / ||u0_b||  (LpLoss(size_average=True).rel).
 * forward leaves the derived fields and per-sample coefficients in `ws` (fno_pino_loss_workspace_bytes);
 * backward writes du = g_ic * dloss_ic/du + g_f * dloss_f/du (device scalars, NULL = 1).
 * n must be 32, 64, 128 (one plane per workgroup, in-LDS radix-2 FFTs) or 256 (three slab passes each way, planes in
 * chunks of 64: csrc/k_pino_loss2.h).
 * ---------------------------------------------------------------------- */
size_t fno_pino_loss_workspace_bytes(int batch, int n, int nt);
int fno_pino_loss_forward(int batch, int n, int nt, const float* u, const float* u0, const float* forcing,
                          const float* visc, float t_interval, float* loss_ic, float* loss_f, void* ws, size_t ws_bytes,
                          void* stream);
int fno_pino_loss_backward(int batch, int n, int nt, const float* u, const float* u0, const float* forcing,
                           const float* visc, float t_interval, const float* g_ic, const float* g_f, float* du, void* ws,
                           size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Gates of the recurrent neural operator cell, neuralop/models/rno.py:254-260 (all tensors (B, C, X, Y) fp32,
 * n elements, 16-byte aligned, n % 4 == 0; b* are the cell's SCALAR biases on the device):
 *   reset gate : r = sigmoid(a3 + a4 + b2), rh = r * h                                   (:256-257)
 *   output gate: z = sigmoid(a1 + a2 + b1), z2 = sigmoid(a7 + a8 + b4), s3 = a5 + a6 + b3,
 *                h_new = (1 - z) * h + z2 * selu(s3)                                     (:254-255, :257-260)
 * a_i = f_i(.) are the cell's Fourier layers.  backward returns the gradient of each pre-activation sum
 * (shared by its two addends), the direct gradient to h, and fno_rno_gate_partials() per-workgroup partial
 * sums of every bias gradient (reset: [P]; output: [3][P] for b1, b4, b3), accumulated and stored in DOUBLE
 * (tens of millions of terms of either sign per scalar), to be summed by the caller.
 * ---------------------------------------------------------------------- */
int fno_rno_gate_partials(void);
int fno_rno_reset_gate_forward(size_t n, const float* a3, const float* a4, const float* b2, const float* h, float* r,
                               float* rh, void* stream);
int fno_rno_reset_gate_backward(size_t n, const float* d_rh, const float* r, const float* h, float* d_s, float* d_h,
                                double* db_partials, void* stream);
int fno_rno_output_gate_forward(size_t n, const float* a1, const float* a2, const float* b1, const float* a7,
                                const float* a8, const float* b4, const float* a5, const float* a6, const float* b3,
                                const float* h, float* z, float* z2, float* s3, float* h_new, void* stream);
int fno_rno_output_gate_backward(size_t n, const float* g, const float* z, const float* z2, const float* s3,
                                 const float* h, float* d_s1, float* d_s7, float* d_s3, float* d_h, double* db_partials,
                                 void* stream);

/* ------------------------------------------------------------------------
 * Pointwise channel mix  y[b, o, p] = sum_i w[o, i] x[b, i, p] + bias[o] + addend[b, o, p]  and its autograd:
 * the Conv1d(k=1) beside every spectral convolution of the observer models
 * (libs/models/pino_models/pinobserver.py:221-226 `sp_convs[i](x) + ws[i](x)`; neuralop/models/rno.py:224-228),
 * for grids the fused block stacks do not cover (odd last dimension).  x, y, addend, dy, dx: (B, C, PW) fp32,
 * C in {32, 64}, PW % 128 == 0; w (C, C) row-major [o][i]; bias / addend / dx_addend / dx / dbias nullable.
 * The gradient w.r.t. `addend` is dy itself.
 * input_gelu = 1: x is a PRE-activation tensor and the mix acts on gelu(x) (the previous layer's activation applied on
 * load, as in the fused block stack); backward then returns dx = (W^T dy + dx_addend) * gelu'(x), where dx_addend is
 * the gradient arriving at gelu(x) from its other consumer (the spectral convolution beside the mix): one layer of
 * `act(sp_conv(x) + w(x))` chains (pinobserver.py:221-226) costs no separate activation or accumulation pass.
 * ---------------------------------------------------------------------- */
size_t fno_pointwise_workspace_bytes(int channels);
int fno_pointwise_forward(int batch, int channels, size_t plane, const float* x, const float* w, const float* bias,
                          const float* addend, int input_gelu, float* y, void* stream);
int fno_pointwise_backward(int batch, int channels, size_t plane, const float* x, const float* w, const float* dy,
                           const float* dx_addend, int input_gelu, float* dx, float* dw, float* dbias, void* ws,
                           size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Projection head on its own:  y = W2 gelu(W1 x + b1) + b2, x (B, C, PW), C in {32, 64}, hidden 128 or 256,
 * Cout = 1, PW % 128 == 0: the `fc1 -> act -> fc2` tail of the observer models
 * (libs/models/pino_models/pinobserver.py:231-233, 270-273; neuralop/models/tfno.py:23-38), with the FNO
 * projection kernels (hidden tensor never materialised; backward recomputes it).  w1 (hidden, C), w2 (1, hidden).
 * backward writes dx and all four parameter gradients.  Needs the split-precision GEMM mode (default).
 * ---------------------------------------------------------------------- */
size_t fno_projection_workspace_bytes(int channels, int hidden);
int fno_projection_forward(int batch, int channels, int hidden, int cout, size_t plane, const float* x, const float* w1,
                           const float* b1, const float* w2, const float* b2, float* y, void* stream);
int fno_projection_backward(int batch, int channels, int hidden, int cout, size_t plane, const float* x,
                            const float* w1, const float* b1, const float* w2, const float* dy, float* dx, float* dw1,
                            float* db1, float* dw2, float* db2, void* ws, size_t ws_bytes, void* stream);
/* The same head with the hidden activation chosen by the caller: FNO_ACT_GELU (the two calls above) or FNO_ACT_RELU,
 * the `Linear(freq_dim, 4 freq_dim) -> ReLU -> Linear(4 freq_dim, 1)` regressor that ends RNO2d
 * (neuralop/models/rno.py:171-175, built with activation='relu' at :319-320; hidden width 256 only). */
#define FNO_ACT_GELU 0
#define FNO_ACT_RELU 1
int fno_projection_forward_act(int batch, int channels, int hidden, int cout, size_t plane, const float* x,
                               const float* w1, const float* b1, const float* w2, const float* b2, int hidden_act,
                               float* y, void* stream);
int fno_projection_backward_act(int batch, int channels, int hidden, int cout, size_t plane, const float* x,
                                const float* w1, const float* b1, const float* w2, const float* dy, int hidden_act,
                                float* dx, float* dw1, float* db1, float* dw2, float* db2, void* ws, size_t ws_bytes,
                                void* stream);

/* ------------------------------------------------------------------------
 * Lifting layer on its own:  y = W x + b,  x (B, Cin <= 4, PW) -> y (B, C, PW), C in {32, 64}, PW % 128 == 0
 * (neuralop/models/tfno.py:11-20; also the `fc0` + Re-conditioning front of the PINO observers,
 * libs/models/pino_models/pinobserver.py:205-207, once its two linear maps are composed).  backward gives the
 * parameter gradients only (x is data).
 * ---------------------------------------------------------------------- */
size_t fno_lifting_workspace_bytes(int channels);
int fno_lifting_forward(int batch, int cin, int channels, size_t plane, const float* x, const float* w, const float* bias,
                        float* y, void* stream);
int fno_lifting_backward(int batch, int cin, int channels, size_t plane, const float* x, const float* dy, float* dw,
                         float* dbias, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Channel-flow Navier-Stokes right-hand side on the staggered grid and the physics-informed loss built on it:
 * NSControlEnvMatlab.compute_rhs_py (libs/envs/control_env.py:429-530) and NSControlEnvMatlab.pde_loss (:627-633),
 * the `pde_loss_weight` branch of the observer training loop (run_pde_observers.py:226-231).
 * One field: U, W (Nx, Ny+1, Nz), V (Nx, Ny, Nz), z contiguous; all calls take `batch` stacked fields.
 *   fno_chanflow_pack_metrics: HOST helper.  From the wall-normal grid y[Ny] (faces), ym[Ny-1] (centres),
 *     yg[Ny+1] (ghost-extended centres, control_env.py:165) fills packed[3*(Ny+2)] with the reciprocal spacings the
 *     kernels index; the caller copies `packed` (doubles) to the device once and passes it as `metrics`.
 *   fno_chanflow_rhs: Fu, Fv, Fw = compute_rhs_py(U, V, W, dPdx); dtype 0 = fp32, 1 = fp64 (the reference's RK3 stepper
 *     runs it in fp64, the training loop on fp32 fields); dpdx = per-sample device array of that dtype or NULL for
 *     `dpdx_default`.
 *   fno_chanflow_pde_loss_forward: loss = sum_b ||Fu(U,Vgt,W)-Fu(U,V,W)|| + ||Fv..|| + ||Fw..|| (fp32, device scalar);
 *     leaves the difference fields and per-sample norms in `ws`.
 *   fno_chanflow_pde_loss_backward: dV = gloss * dloss/dV (gloss: device scalar, NULL = 1); Vgt is data.
 * ---------------------------------------------------------------------- */
typedef struct FnoChanflowGrid {
  int Nx, Ny, Nz;        /* Ny = number of y faces (rows of V); U and W carry Ny+1 rows */
  double dx, dz, nu;
} FnoChanflowGrid;
int fno_chanflow_pack_metrics(int Ny, const double* y, const double* ym, const double* yg, double* packed);
int fno_chanflow_rhs(const FnoChanflowGrid* grid, int batch, int dtype, const double* metrics, const void* U, const void* V,
                     const void* W, const void* dpdx, double dpdx_default, void* Fu, void* Fv, void* Fw, void* stream);
size_t fno_chanflow_pde_loss_workspace_bytes(const FnoChanflowGrid* grid, int batch);
int fno_chanflow_pde_loss_forward(const FnoChanflowGrid* grid, int batch, const double* metrics, const float* U,
                                  const float* Vgt, const float* V, const float* W, float* loss, void* ws, size_t ws_bytes,
                                  void* stream);
int fno_chanflow_pde_loss_backward(const FnoChanflowGrid* grid, int batch, const double* metrics, const float* U,
                                   const float* Vgt, const float* V, const float* W, const float* gloss, float* dV, void* ws,
                                   size_t ws_bytes, void* stream);

/* Names and average device time (ms, HIP events on `stream`) of the kernels launched
 * by the last fno_model_* call made with profiling enabled; used by bench.py for the
 * roofline line.  fno_profile_enable(1) makes every launch event-bracketed (slow path). */
void fno_profile_enable(int on);
int fno_profile_count(void);
int fno_profile_get(int idx, const char** name, float* total_ms, int* launches);
/* which matrix pipe record idx's channel GEMMs ran on: 0 not stated, 1 fp32 MFMA, 2 two fp16 terms per operand (3 products per
 * k block), 3 three bf16 terms (6 products) - what a roofline has to price the kernel against */
int fno_profile_get_terms(int idx);
void fno_profile_reset(void);

#ifdef __cplusplus
}
#endif
#endif
