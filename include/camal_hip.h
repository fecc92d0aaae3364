/*
 * camal_hip.h — C ABI of libcamal_hip.so: CamAL's detection + CAM-localization hot path
 * (and the training step of the same ResNets) as hand-written HIP kernels for gfx950.
 *
 * The reference (adrienpetralia/CamAL) has no FFI of its own: its boundary is the Python
 * surface of `CamAL` / `CamALResNet` / `CAM`.  Each entry point below names
 * the reference code (file:line under the reference root) whose arithmetic it replaces; the
 * Python mirror in camal_amd/ keeps the reference's names, arguments and errors and calls
 * these through ctypes (see INTEGRATION.md for the binding).
 *
 * Conventions
 *   - every function returns 0 on success or a negative CAMAL_ERR_* code; the message is in
 *     camal_last_error() (thread-local, valid until the next failing call on that thread);
 *   - `*_dev` pointers are DEVICE pointers BORROWED from the caller (e.g. torch tensors'
 *     data_ptr()); the library never frees them.  The context owns only its packed weights
 *     and its activation workspace;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work is enqueued
 *     on it and nothing synchronises, except camal_create / set_member_weights / destroy;
 *   - one context per (device, stream of use); no internal locking; no exceptions cross
 *     the ABI.
 */
#ifndef CAMAL_HIP_H
#define CAMAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAMAL_OK 0
#define CAMAL_ERR_ARG (-1)     /* bad argument (EINVAL-like) */
#define CAMAL_ERR_HIP (-2)     /* a HIP runtime call failed */
#define CAMAL_ERR_STATE (-3)   /* e.g. member weights not loaded */
#define CAMAL_ERR_NOMEM (-4)

#define CAMAL_FP32 0 /* fp32 activations, v_mfma_f32_32x32x2_f32 (exact fp32)            */
#define CAMAL_BF16 1 /* bf16 activations/weights, v_mfma_f32_16x16x32_bf16, fp32 accumulate */

/* Classifier architectures (same state-dict key names, widths 1 -> 64 -> 128 -> 128, same head and CAM):       */
#define CAMAL_ARCH_CAMAL 0  /* CamALResNet, camal_resnet.py:4-89: kernels (k, 7, 3) per block, replicate padding   */
#define CAMAL_ARCH_RESNET 1 /* ResNet, classifiers/resnet.py:3-68: kernels (7, 5, 3) per block, zero "same" padding */

typedef struct camal_ctx camal_ctx;

/* Version / build info of the library ("camal_hip <n> gfx950"). */
const char* camal_version(void);
const char* camal_last_error(void);

/* Number of floats in the folded-weight blob of one member with first-kernel size k.
 * Blob layout (all float32, BatchNorm already folded into conv weight/bias):
 *   for each of the 9 ConvBlocks in model order (camal_resnet.py:13-18,35-38):
 *       w'[Cout][Cin][k_i], b'[Cout]           (k_i = k,7,3 per block; camal_resnet.py:31)
 *   residual of block 1: wr'[64][1],  br'[64]  (camal_resnet.py:51-54)
 *   residual of block 2: wr'[128][64], br'[128]
 *   head: W_fc[2][128], b_fc[2]                (camal_resnet.py:19)                      */
size_t camal_member_blob_floats(int kernel_size);

/* Replaces CamAL.eval()'s model materialisation (core.py:366-413): one context holds the
 * E ensemble members (kernel_sizes[E], each odd, 1..63) on `device_id`.                   */
int camal_create(int device_id, int n_members, const int* kernel_sizes, int precision, camal_ctx** out);
int camal_destroy(camal_ctx* ctx);

/* camal_create for either classifier.  CAMAL_ARCH_RESNET replaces instantiating classifiers/resnet.py's ResNet
 * (resnet.py:3-23; every kernel_sizes[e] must be 7, its first kernel): blob layout as above with k_i = 7,5,3,
 * layers padded with zeros (resnet.py:58-62).  camal_set_fused applies to it like to CamALResNet.             */
int camal_create_arch(int device_id, int n_members, const int* kernel_sizes, int precision, int arch, camal_ctx** out);
size_t camal_member_blob_floats_arch(int kernel_size, int arch); /* 0 for an unknown arch */

/* The same for CamALResNet(mid_channels=...) (camal_resnet.py:8,14-16: widths 1 -> mid -> 2 mid -> 2 mid; what
 * CamAL.train builds from its resnet_kwargs, core.py:36-38,106).  mid_channels 32, 64 or 128; 64 is camal_create_arch.
 * Other widths run one kernel per ConvBlock, in fp32 or bf16 (the fused-block kernels are built for 64).  Blob layout
 * as above with 64 / 128 replaced by mid / 2 mid.                                                                 */
int camal_create_widths(int device_id, int n_members, const int* kernel_sizes, int precision, int arch, int mid_channels,
                        camal_ctx** out);
size_t camal_member_blob_floats_widths(int kernel_size, int arch, int mid_channels); /* 0 if unsupported */

/* Upload one member's folded weights (HOST pointer, layout above); the library re-lays them
 * out into MFMA fragment order.  Replaces load_state_dict + .to(device) (core.py:391-398). */
int camal_set_member_weights(camal_ctx* ctx, int member, const float* folded_host, size_t n_floats);

/* Windows processed per internal pass (the activation workspace is sized for this many windows).
 * Default when never called: about 2 M timesteps per pass, at most 1024 windows (measured optimum:
 * 1024 windows of 1440 steps, 256 of 10080).                                                */
int camal_set_chunk_windows(camal_ctx* ctx, int windows);

/* Size the context's workspace now for calls of up to B windows of L steps (camal_forward_cam, camal_localize), so
 * that those calls allocate nothing later: without it the first call of a larger shape frees and re-allocates the
 * workspace on the call path (not capturable in a HIP graph, and a first-call latency of its own).            */
int camal_reserve(camal_ctx* ctx, int B, int L);

/* 1 (default for bf16; fp32 defaults to 0; env CAMAL_FUSED overrides): blocks 2 and 3 of every member run as fused-block kernels (three ConvBlocks per
 * launch, intermediates in LDS); 0: one kernel per ConvBlock.  Same arithmetic per element; the
 * pooled sums are taken over different tile sizes, so logits agree to fp32 rounding.          */
int camal_set_fused(camal_ctx* ctx, int fused);

/* ResNet forward + CAM for every member.  Replaces CamALResNet.forward
 * (camal_resnet.py:21-23) and CAM.run's device part (class_activation_map.py:51-74:
 * forward, SaveFeatures hook, W_fc[label]·F with label=1) for a batch of windows.
 *   x_dev      [B][L] float32 (the model input, i.e. the scaled aggregate)
 *   logits_dev [E][B][2] float32 out
 *   cam_dev    [E][B][L] float32 out, or NULL.  cam[t] = sum_c W_fc[cam_label][c] * F[c][t]
 *   member     -1 = all members, else only that member (outputs then have E = 1)
 *   cam_label  0 or 1: `returned_cam_for_label` of CAM.run (class_activation_map.py:34,72)  */
int camal_forward_cam(camal_ctx* ctx, const float* x_dev, int B, int L, int member, int cam_label,
                      float* logits_dev, float* cam_dev, void* stream);

/* camal_forward_cam with the hook on another layer (CAM's last_conv_layer argument, class_activation_map.py:25-31):
 * tap_block 2 = the last ResNetBlock (camal_forward_cam, what CamAL.eval passes, core.py:403); 1 = the second
 * ResNetBlock, whose output has the width the head's weights have: cam[t] = sum_c W_fc[cam_label][c] * F2[c][t].
 * The logits are the model's either way.  (The first block's output is narrower than W_fc: the reference's dot
 * product fails on it.)                                                                                           */
int camal_forward_cam_tap(camal_ctx* ctx, const float* x_dev, int B, int L, int member, int cam_label, int tap_block,
                          float* logits_dev, float* cam_dev, void* stream);

/* Fused detection + localization + power.  Replaces the body of CamAL.predict
 * (core.py:286-327: member loop, softmax, clip/normalise/average, vote gate,
 * attention_sigmoid_module :415-437) and CamAL.test's power step (core.py:216-218).
 *   x_dev       [B][L] float32 model input
 *   x64_dev     [B][L] float64 copy of the caller's data used for the attention product
 *               (core.py:434 multiplies by the un-cast `data`), or NULL (then x_dev is used)
 *   y_dev       [B] int8 optional per-window gate (`y` of predict, core.py:297-300), or NULL
 *   w_ma        moving-average window (core.py:257), >= 1
 *   gate_mode   0: reference device=="cpu" path (no pre-filter)
 *               1: reference device!="cpu" path: member 0's label gates every member
 *                  (core.py:292-300).  Ignored when y_dev != NULL.
 *   pred_f64    0: ensemble CAM accumulates in float32 (caller's data was float32)
 *               1: in float64 (caller's data was float64) — core.py:286 zeros_like(data)
 *   p_on        appliance mean on-power (core.py:216); used only if power_out != NULL
 *   scale_mul, scale_add   inverse scaling agg = x*scale_mul + scale_add
 *               (torch_dataset.py:131-156)
 *   prob_out    [B] float64 out: ensemble detection probability (core.py:320)
 *   avg_cam_out [B][L] float32 out or NULL: vote-gated average CAM before attention (:324)
 *   state_out   [B][L] float32 out: on/off states (values -1,-0,0,1 like np.round, :437)
 *   power_out   [B][L] float32 out or NULL: min(state*p_on, agg)  (core.py:216-218)
 *   member_logits_out [E][B][2] / member_cam_out [E][B][L]: optional per-member outputs    */
int camal_localize(camal_ctx* ctx, const float* x_dev, const double* x64_dev, const int8_t* y_dev,
                   int B, int L, int w_ma, int gate_mode, int pred_f64,
                   double p_on, double scale_mul, double scale_add,
                   double* prob_out, float* avg_cam_out, float* state_out, float* power_out,
                   float* member_logits_out, float* member_cam_out, void* stream);

/* camal_localize with the states as ONE BYTE per timestamp, for callers that ship results to the host
 * (CamAL.predict's NumPy-out contract, core.py:329-332, at configs[2]'s 10 M windows: 1 byte instead of 4 over PCIe and
 * in the pinned staging buffers; the host widens to the caller's dtype).  np.round's four values (core.py:437) as
 * int8: 0, 1, -1, and CAMAL_STATE_NEG_ZERO for -0.0.  Same arithmetic and the same decisions as camal_localize
 * (one kernel, two store forms); arguments as there.                                                              */
#define CAMAL_STATE_NEG_ZERO (-128)
int camal_localize_i8(camal_ctx* ctx, const float* x_dev, const double* x64_dev, const int8_t* y_dev,
                      int B, int L, int w_ma, int gate_mode, int pred_f64,
                      double p_on, double scale_mul, double scale_add,
                      double* prob_out, int8_t* state_out, float* power_out, void* stream);

/* Post-processing only, on CAM rows / logits the caller already holds on the device
 * (same arithmetic as the tail of camal_localize).  cam_dev [E][B][L], logits_dev [E][B][2]. */
int camal_postprocess(int n_members, const float* cam_dev, const float* logits_dev,
                      const float* x_dev, const double* x64_dev, const int8_t* y_dev,
                      int B, int L, int w_ma, int gate_mode, int pred_f64,
                      double p_on, double scale_mul, double scale_add,
                      double* prob_out, float* avg_cam_out, float* state_out, float* power_out,
                      void* stream);

/* CamAL.attention_sigmoid_module on its own (core.py:415-437): moving average (zero edges),
 * rectified sigmoid of ma*x, round half to even.  soft_dev, x_dev, state_out: [B][L] float64. */
int camal_attention(const double* soft_dev, const double* x_dev, int B, int L, int w_ma,
                    double* state_out, void* stream);

/* Reductions behind NILMmetrics (src/helpers/other.py:112-159) as CamAL.test applies them to the
 * flattened states and powers (core.py:199-231), so that [N][L] results never round-trip to the host.
 * All arrays are DEVICE pointers of n elements:
 *   y32_dev / y64_dev  true appliance power as float32 or float64 (one of them, or both NULL to skip
 *                      the regression sums); yhat_dev estimated power (camal_localize power_out);
 *   ystate_dev         true on/off labels as int8 (0/1); yhat_state_dev estimated states
 *                      (camal_localize state_out); both NULL to skip the confusion counts.
 *   sums_out           12 doubles: sum|yh-y|, sum(yh-y)^2, sum|y|, sum y^2, sum y, sum yh, sum min(yh,y),
 *                      sum max(yh,y) (after nan_to_num of y, yh: nan/inf -> 0), TP, TN, FP, FN.
 *   workspace_dev      scratch of camal_nilm_workspace_bytes(n) bytes (per-block partials; the library
 *                      allocates nothing).  Deterministic: fixed two-stage summation order.
 * MAE = s0/n, MSE = s1/n, TECA = 1 - s0/(2 s2), NDE = s1/s3, SAE = |s5-s4|/s4, MR = s6/s7.           */
#define CAMAL_NILM_NSUMS 12
size_t camal_nilm_workspace_bytes(size_t n);
int camal_nilm_sums(const float* y32_dev, const double* y64_dev, const float* yhat_dev, const int8_t* ystate_dev,
                    const float* yhat_state_dev, size_t n, double* sums_out, void* workspace_dev,
                    size_t workspace_bytes, void* stream);

/* Window builder + scaler + weak labels in front of camal_localize, on the device (the data that feeds the hot
 * path never takes the pandas / NumPy detour): replaces the window loop of *_DataBuilder.get_nilm_dataset
 * (src/helpers/data_processing.py:377-400), NILMscaler.transform on the aggregate channel
 * (src/helpers/torch_dataset.py:105-111) and nilmdataset_to_clfdataset (data_processing.py:17-33).
 *   series_dev [n_rows][T]  float64 (series_is_f64 = 1) or float32 rows: row 0 aggregate power, then (power,
 *                           status) per appliance, as _get_stems lays them out.
 *   windows i = 0 .. camal_windows_count(T, W, stride) - 1 start at i * stride; a window with a NaN in ANY row
 *   is dropped (:388); the survivors keep their order.  Outputs are sized for all windows, the first
 *   *count_out (device int) rows are written:
 *   x_out [n][W]        float32((aggregate - stat1) / stat2), the division in float64 as NILMscaler does it
 *                       ('MinMax': pass stat2 = max - min); this is what CamALResNet is fed (.float());
 *   x64_out [n][W]      the same in float64 (what predict's attention module multiplies by), or NULL;
 *   label_out [n]       int8 weak label: any(status of appliance `app` != 0) (data_processing.py:28), or NULL;
 *   app_power_out, app_state_out [n][W]  float32 power / int8 status rows of appliance `app` (the y and
 *                       y_state of CamAL.test's metrics), or NULL;
 *   index_out [n]       int32 index i of each surviving window, or NULL.
 *   workspace_dev       camal_windows_workspace_bytes(n_wins) bytes; the library allocates nothing.         */
int camal_windows_count(long long T, int W, int stride);
size_t camal_windows_workspace_bytes(int n_wins);
int camal_build_windows(const void* series_dev, int series_is_f64, int n_rows, long long T, int W, int stride, int app,
                        double stat1, double stat2, float* x_out, double* x64_out, int8_t* label_out,
                        float* app_power_out, int8_t* app_state_out, int* index_out, int* count_out,
                        void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same window builder for callers that cannot allocate or pass a workspace (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference: it exists only for
 * those callers.  The scratch (NaN flags, scan positions, block totals) lives in a file-scope device array of
 * camal_windows_static_scratch_bytes() = 5 MiB + 4 KiB that every GPU the library runs on holds until the process
 * ends; the call runs on the GPU that holds series_dev and x_out (taken from the pointers: CAMAL_ERR_ARG when they
 * are not device memory of ONE GPU; the caller's current device is put back), makes no hipMalloc / hipFree and
 * returns CAMAL_ERR_ARG when the series yields more than CAMAL_WINDOWS_STATIC_MAX windows.  All calls on one GPU share that array: calls that overlap
 * on one GPU (two streams, two threads) race — serialise them.  Arguments and results as camal_build_windows. */
#define CAMAL_WINDOWS_STATIC_MAX (1 << 20)
size_t camal_windows_static_scratch_bytes(void);
int camal_build_windows_static(const void* series_dev, int series_is_f64, int n_rows, long long T, int W, int stride,
                               int app, double stat1, double stat2, float* x_out, double* x64_out, int8_t* label_out,
                               float* app_power_out, int8_t* app_state_out, int* index_out, int* count_out,
                               void* stream);

/* ------------------------------- training (one ResNet member) -------------------------------
 * Replaces the per-batch body of BasedClassifTrainer.__train (torch_trainer.py:281-298) for
 * CamALResNet in train() mode: forward with BatchNorm batch statistics (biased variance,
 * running statistics updated with momentum and the unbiased variance), mean cross-entropy,
 * backward, AdamW.  Parameters, gradients, optimizer moments and BN running statistics are
 * flat fp32 DEVICE buffers owned by the caller (torch tensors), so that the data-parallel
 * gradient all-reduce is one torch.distributed call on the gradient bucket between
 * camal_train_fwd_bwd and camal_adamw_step (replaces nn.DataParallel, torch_trainer.py:82-90).
 *
 * Flat parameter order = CamALResNet.parameters() order (camal_resnet.py:13-19,35-54): for each
 * block: 3 x (conv.weight [Cout][Cin][k], conv.bias, bn.weight, bn.bias), then the residual's
 * (conv.weight [Cout][Cin][1], conv.bias, bn.weight, bn.bias) for blocks 1 and 2; finally
 * linear.weight [2][128], linear.bias [2].  Running-statistics order: per BatchNorm in the same
 * order, running_mean [C] then running_var [C].                                              */
typedef struct camal_trainer camal_trainer;
size_t camal_trainer_param_count(int kernel_size);   /* number of trainable floats P(k)        */
size_t camal_trainer_running_count(void);            /* floats of BN running statistics        */
int camal_trainer_create(int device_id, int kernel_size, int max_batch, int max_len, camal_trainer** out);
/* The same for either classifier (CAMAL_ARCH_RESNET: classifiers/resnet.py's ResNet trained by the same
 * BasedClassifTrainer.__train body, torch_trainer.py:273-300; kernel_size must be 7): parameter order = that model's
 * model.parameters(), layers read zeros outside the window, so the data gradient needs no edge fold.            */
int camal_trainer_create_arch(int device_id, int kernel_size, int max_batch, int max_len, int arch, camal_trainer** out);
size_t camal_trainer_param_count_arch(int kernel_size, int arch); /* 0 for an unknown arch */
/* The same for mid_channels 32, 64 or 128 (CamAL.train with resnet_kwargs, core.py:36-38,106). */
int camal_trainer_create_widths(int device_id, int kernel_size, int max_batch, int max_len, int arch, int mid_channels,
                                camal_trainer** out);
size_t camal_trainer_param_count_widths(int kernel_size, int arch, int mid_channels); /* 0 if unsupported */
size_t camal_trainer_running_count_widths(int mid_channels);                          /* 0 if unsupported */
int camal_trainer_destroy(camal_trainer* t);

/* Data-parallel training (replaces nn.DataParallel, torch_trainer.py:82-90, whose loss is the mean over the
 * GATHERED batch, :288-291): a rank that holds b of a step's B samples sets scale = b / B; camal_train_fwd_bwd then
 * writes scale x the gradient of its own mean cross-entropy, so that ONE sum all-reduce of the ranks' flat gradient
 * buckets is the gradient of the global mean loss, ragged shards included.  loss_dev stays the rank's unscaled mean.
 * Default 1 (single-GPU training).                                                                              */
int camal_trainer_set_grad_scale(camal_trainer* t, float scale);

/* One forward(+backward) pass on a batch.
 *   params_dev      flat parameters (read)
 *   bn_running_dev  flat running statistics, updated in place (NULL: not updated)
 *   x_dev [B][L] float32, y_dev [B] int32 labels (may be NULL when grads_dev is NULL)
 *   bn_momentum     0.1 in the reference (torch default)
 *   grads_dev       flat gradients of the MEAN cross-entropy, overwritten (NULL: forward only)
 *   logits_dev      [B][2] out or NULL; loss_dev: 1 float out; ncorrect_dev: 1 int out
 * Conv biases feed a train-mode BatchNorm, so their exact gradient is 0 and is written as 0
 * (torch produces rounding noise there).                                                      */
int camal_train_fwd_bwd(camal_trainer* t, const float* params_dev, float* bn_running_dev,
                        const float* x_dev, const int* y_dev, int B, int L, float bn_momentum,
                        float* grads_dev, float* logits_dev, float* loss_dev, int* ncorrect_dev,
                        void* stream);

/* Backward pass for a caller-supplied training criterion (BasedClassifTrainer's `criterion` argument,
 * torch_trainer.py:20,52,290-291: any callable on (logits, labels)).  Protocol for one batch:
 *   1. camal_train_fwd_bwd(t, params, running, x, NULL, B, L, momentum, NULL (= grads), logits_dev, NULL, NULL, stream)
 *      = the train-mode forward; the activations stay in the trainer's workspace;
 *   2. the caller evaluates its loss on logits_dev and differentiates it with respect to them (torch autograd on a
 *      [B][2] tensor) -> dlogits_dev [B][2] float32;
 *   3. camal_train_bwd_dlogits: the head's and every layer's gradients from dlogits_dev (x the grad scale), written to
 *      grads_dev exactly as camal_train_fwd_bwd writes them for the cross-entropy.
 * y_dev: labels for the accuracy count (NULL: ncorrect is meaningless); loss_dev receives the batch's mean
 * cross-entropy when y_dev is given (the caller's own loss value is the caller's).  No other call on this trainer
 * may come between steps 1 and 3.                                                                                */
int camal_train_bwd_dlogits(camal_trainer* t, const float* params_dev, const float* x_dev, const int* y_dev,
                            const float* dlogits_dev, int B, int L, float* grads_dev, float* loss_dev, int* ncorrect_dev,
                            void* stream);

/* The activations the last camal_train_fwd_bwd left in the trainer's workspace, for layer-level parity tests (what a
 * forward hook on a module shows in the reference, class_activation_map.py:8-12 being the one it registers itself):
 *   what = CAMAL_PEEK_Z       pre-BatchNorm conv output of slot `index` [B][L][C]; slots in model.parameters() order:
 *                             block 1: conv 0, 1, 2, skip conv; block 2: conv 0, 1, 2, skip conv; block 3: conv 0, 1, 2
 *          CAMAL_PEEK_ACT     ConvBlock / block outputs 0..8 [B][L][C] (channels-last)
 *          CAMAL_PEEK_MEAN / CAMAL_PEEK_INVSTD   batch statistics of slot `index` [C]
 * Copies B*L*C (Z, ACT) or C (MEAN, INVSTD) floats into out_dev on `stream` (out_floats = its capacity; out_dev
 * NULL: only *channels_out = C is returned).  B, L: the shape of that last call.                                    */
#define CAMAL_PEEK_Z 0
#define CAMAL_PEEK_ACT 1
#define CAMAL_PEEK_MEAN 2
#define CAMAL_PEEK_INVSTD 3
int camal_trainer_peek(camal_trainer* t, int what, int index, int B, int L, float* out_dev, size_t out_floats,
                       int* channels_out, void* stream);

/* Evaluation loss of BasedClassifTrainer.evaluate / __evaluate (torch_trainer.py:188-193,316-321) from eval-mode
 * logits (camal_forward_cam): per-window cross-entropy (float32, -log_softmax[label]) and predicted label (argmax,
 * the first maximum like torch.max).  logits_dev [B][2] float32, labels_dev [B] int32, loss_out [B] float32,
 * pred_out [B] int32; the caller forms the per-batch means the reference logs.                                    */
int camal_ce_eval(const float* logits_dev, const int* labels_dev, int B, float* loss_out, int* pred_out, void* stream);

/* torch.optim.AdamW update of a flat buffer (torch_trainer.py:35,292): decoupled weight decay,
 * bias-corrected moments; `step` counts from 1.                                                */
int camal_adamw_step(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, size_t n,
                     int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                     void* stream);

/* HIP-event timing of the kernels the last camal_forward_cam / camal_localize call enqueued
 * (enable with camal_set_profiling(ctx, 1); costs two events per launch).  After the stream
 * has been synchronised, camal_get_profile returns, per kernel class, the number of launches
 * and the summed duration in milliseconds.  Classes: 0 stem, 1 conv (MFMA implicit GEMM),
 * 2 conv+CAM head, 3 post-processing, 4 other.                                             */
#define CAMAL_PROF_CLASSES 5
int camal_set_profiling(camal_ctx* ctx, int enabled);
int camal_get_profile(camal_ctx* ctx, int* launches /*[5]*/, double* ms /*[5]*/);

#ifdef __cplusplus
}
#endif
#endif /* CAMAL_HIP_H */
