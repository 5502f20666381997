/*
 * rced.h -- C ABI of the MI355X-native R-CED / CR-CED forward pass (librced_hip.so).
 *
 * This is the drop-in boundary for the one hot path of phecda-xu/FullyCNNSpeechEnhancement:
 * what the reference runs as `sess.run(self.pred, {self.input_x: x})`
 *   (model_utils/tester.py:85-90, infer.py:62-65, model_utils/trainer.py:245-250)
 * over the graph built by `self.pred = self.model(self.input_x)`
 *   (model_utils/tester.py:69-83; model_utils/model.py:6-96; model_utils/module.py:11-34).
 *
 * Plain C types only; integer status codes; caller-owned buffers; no exceptions cross the
 * boundary.  A model handle is NOT thread-safe (one stream per call, like the reference's
 * single tf.Session).  There is no CPU fallback: every entry point that computes needs a
 * gfx950 device and fails with RCED_ERR_HIP otherwise.
 *
 * Tensor layout (module.py:15, data_loader.py:206-208): NHWC float32,
 *   x, y : [N, T, 129, 1]   N utterances, T time frames, 129 frequency bins.
 *
 * Weight blob (float32), per layer in graph order -- exactly the TF variables of
 * module.py:27,29, raw (BatchNorm is folded inside rced_create, not by the caller):
 *   "{scope}/kernel"                      [kh, kw, cin, cout]  (HWIO)
 *   "{scope}/bias"                        [cout]
 *   "{scope}/batch_norm/gamma"            [cout]   } only for layers with use_norm
 *   "{scope}/batch_norm/beta"             [cout]   }
 *   "{scope}/batch_norm/moving_mean"      [cout]   }
 *   "{scope}/batch_norm/moving_variance"  [cout]   }
 */
#ifndef RCED_H_
#define RCED_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCED_FEATURE_DIM 129 /* cfg [data] feature_dim (Work/.../fully_cnn_*.cfg) */

/* net_work selection of infer.py:45-51 / tester.py:76-82 */
#define RCED_V1 1 /* FullyCNNSEModel    R-CED 10 layers  model.py:6-29  */
#define RCED_V2 2 /* FullyCNNSEModelV2  R-CED 16 layers  model.py:32-61 */
#define RCED_V3 3 /* FullyCNNSEModelV3  CR-CED 16 layers model.py:64-96 */

/* status codes */
#define RCED_OK 0
#define RCED_ERR_ARG 1    /* bad argument (shape, null pointer, unknown variant)  */
#define RCED_ERR_HIP 2    /* HIP runtime error or no gfx950 device                */
#define RCED_ERR_ALLOC 3  /* device or host allocation failed                     */
#define RCED_ERR_STATE 4  /* handle used after destroy / wrong device             */

/* execution paths (rced_set_option "path") */
#define RCED_PATH_AUTO 0      /* fused kernels where available, else layerwise      */
#define RCED_PATH_LAYERWISE 1 /* one generic direct-conv launch per layer           */
#define RCED_PATH_FUSED 2     /* fused multi-layer MFMA kernels; error if unavailable */

typedef struct rced_model rced_model;

/* Topology queries (host only; no device needed).  Replace reading model.py by hand. */
int rced_num_layers(int variant);                 /* 10 / 16 / 16, or -1 */
size_t rced_num_weights(int variant);             /* blob length in floats, or 0 */
size_t rced_num_trainable(int variant);           /* 32765 / 32192 / 32653 (readme.md:65-67) */
/* Layer i of `variant`: out[0..8] = cout,kh,kw,use_norm,use_act,src,skip_pre,skip_post,cin.
 * src/skip ids: 0 = network input, k+1 = output of layer k, -1 = none. */
int rced_layer_desc(int variant, int layer, int out[9]);
const char* rced_layer_scope(int variant, int layer); /* TF variable scope, e.g. "CE1_encode_1" */

/* Model(is_training=False) + Saver.restore: tester.py:69-83, 36-39.
 * blob: host pointer, n_floats == rced_num_weights(variant).  device: HIP ordinal. */
int rced_create(int variant, const float* blob, size_t n_floats, int device, rced_model** out);
void rced_destroy(rced_model* m);

/* y = model(x).  x_dev / y_dev: DEVICE pointers to [N,T,129,1] float32, resident on the
 * model's device.  stream: a hipStream_t (NULL = default stream).  Asynchronous: returns
 * after enqueueing.  Replaces sess.run(self.pred, ...) with device-resident tensors. */
int rced_forward(rced_model* m, const float* x_dev, float* y_dev, int N, int T, void* stream);

/* State of the model AFTER work the caller has synchronised itself.  rced_forward only enqueues; the fused CR-CED kernel
 * records a wave-to-wave hand-off that timed out in a sticky word in pinned host memory, which the library looks at before
 * the NEXT launch and in its synchronising entry points (rced_forward_host, rced_profile_query) -- a caller that enqueues
 * one forward, synchronises its own stream and reads y calls this to learn that y is valid: RCED_OK, or RCED_ERR_STATE
 * (the masks of that launch are wrong; destroy and recreate the model).  No device call, no synchronisation. */
int rced_check(rced_model* m);

/* Same with HOST pointers (the reference boundary hands numpy arrays: tester.py:85-90).
 * Copies H2D, runs, copies D2H, synchronises.  Batches >= 8 MB are split into "host_chunks" utterance chunks and
 * the three legs are overlapped on internal streams (a helper thread issues the downloads). */
int rced_forward_host(rced_model* m, const float* x_host, float* y_host, int N, int T);

/* Pre-size the internal workspace for shapes up to [N,T,...] so that rced_forward performs
 * no allocation (needed before stream capture into a hipGraph). */
int rced_reserve(rced_model* m, int N, int T);

/* Options (set/get unless noted).  Returns RCED_ERR_ARG for unknown keys/values.
 *   "path"        RCED_PATH_*
 *   "profile"     1: HIP events around every kernel launch (read with rced_profile_query); set re-arms
 *   "host_chunks" pipeline depth of rced_forward_host (0 = default 8, 1 = no overlap, <= 64)
 *   "fused_grid"  workgroups of the persistent fused kernel (0 = one per CU)
 *   "bf16"        R-CED V1 / V2 only: 1 = bf16 activations + inner-layer weights, fp32 accumulation (BASELINE config 2;
 *                 ~6e-3 of the largest output away from the fp32 result -- opt-in)
 *   "bf16_frames" R-CED V1 / V2 in "bf16" mode: frames (= waves) per workgroup of the kernel, 0 (default) = chosen per call -- eight once
 *                 the call has 16 frames per CU, four below that --, 4 or 8 = always that form.  Results are bit-identical either way.
 *   "v3_l2x6"     CR-CED only: which form of the fused kernel runs.  3 (default) = EVERY layer at fp32 quality on the bf16 matrix pipe
 *                 (three-part operands, six products): the first layer from an im2col-along-time of the input rows, the 18 -> 30 and
 *                 30 -> 8 layers as one stream in which the 30-channel tensor never leaves the registers, decode_final as a GEMM over a
 *                 tap table resident in LDS; 0 = every layer on the fp32 MFMA (bit-for-bit an fp32 fmaf chain; the in-build comparator:
 *                 the two agree to ~1e-6 of the largest output).  The intermediate forms of rounds 3 - 4 (1, 2) are in the library only
 *                 when it is compiled with -DRCED_V3_LEGACY_FORMS=1; otherwise they are refused ("v3_l2x6 takes 3 ... or 0").  A form's
 *                 weight stream is built when it is first selected.
 *   "final_x6", "final_lds"   R-CED V1 / V2 only, fp32 mode: the 1x129 output layer's kernel -- three-part bf16 products (1,
 *                 default) or the fp32 MFMA (0), the latter with (1) / without (0) LDS staging of its B operand.  (In "bf16" mode the
 *                 output layer runs inside the one fused kernel.)
 *   "latency_form"  R-CED V1 / V2 only (fp32 kernel): 1 (default) = a call with fewer 3-frame tiles than the part has CUs (BASELINE
 *                 config 1: one utterance of 256 frames) runs on ONE-frame tiles -- three times the workgroups, a third of the work
 *                 each, bit-identical results; 0 = always 3-frame tiles
 *   "inject_handoff_error"  set only, CR-CED: writes the value into the sticky hand-off error word as the kernel would on a
 *                 time-out (0 clears it) -- a test hook for rced_check / RCED_ERR_STATE handling
 *   "has_fused", "num_cus", "fused_final"  get only ("fused_final": the 1x129 output layer runs inside the fused kernel)
 * Options are PER HANDLE.  Environment variables only supply DEFAULTS, read once when a handle is created (rced_create /
 * rced_train_create) and never afterwards: RCED_V3_L2X6, RCED_FINAL_X6, RCED_FINAL_LDS (the options of the
 * same meaning above); for rced_train_create RCED_TRAIN_MFMA=0 (direct-conv kernels only), RCED_TRAIN_FUSE_ACT=0,
 * RCED_TRAIN_FUSE_DZ=0 (materialise activations / dz), RCED_TRAIN_FUSE_SUMS, RCED_TRAIN_FUSE_BWD, RCED_TRAIN_DET, RCED_TRAIN_X6. */
int rced_set_option(rced_model* m, const char* key, int value);
int rced_get_option(rced_model* m, const char* key, int* value);

/* The single op, module.py:11-34 conv_bn_relu with is_training=False, on DEVICE pointers:
 *   x [N,T,F,cin] -> y [N,T,F,cout];  kernel [kh,kw,cin,cout], bias [cout] (device);
 *   bn = gamma,beta,moving_mean,moving_variance (4*cout floats, device) or NULL (use_norm=False);
 *   skip_input [N,T,F,cout] or NULL; use_act 0/1.  padding SAME, stride 1. */
int rced_conv_bn_relu(const float* x, float* y, const float* kernel, const float* bias,
                      const float* bn, const float* skip_input, int use_act, int N, int T, int F,
                      int cin, int cout, int kh, int kw, int device, void* stream);

/* ---- audio front-end / back-end around the CNN, fixed to the reference's
 * configuration: 8 kHz, 256-sample hamming window, 128-sample stride, rfft(256) -> 129 bins. ---- */

/* Frames the reference produces for a signal of `length` samples: ceil(|L-256|/128 + 1)
 * (data_utils/audio_feature.py:70). */
int rced_stft_num_frames(int length);

/* AudioFeature.compute_spectrogram + power_spectrum + divide_phase (audio_feature.py:22-44,102-115) for a
 * batch, laid out as DataLoader.padding_batch does (data_loader.py:198-209):
 *   pcm_dev [N, L] float32; lengths_dev [N] int32 (device) or NULL (= L each);
 *   mag_dev [N, T, 129] float32 (= the CNN's [N, T, 129, 1] input);
 *   phase_dev [N, T, 129, 2] float32 (re, im of exp(j*angle)) or NULL.
 * Frames past an utterance's own rced_stft_num_frames(length) are zero magnitude, phase 1+0j. */
int rced_stft(const float* pcm_dev, const int* lengths_dev, int N, int L, int T, float* mag_dev,
              float* phase_dev, int device, void* stream);

/* AudioReBuild.rebuild_audio (model_utils/utils.py:171-183): merge mag*phase, irfft(n=nfft)[:256],
 * divide by the hamming window, keep the first half of frame 0 and the second half of every frame,
 * de-emphasis.  audio_dev [N, (T+1)*128] float32; the caller trims row n to its signal length.
 * nfft = 512 reproduces the reference as shipped (AudioReBuild() default although the STFT used 256);
 * nfft = 256 is the matching inverse. */
int rced_istft(const float* mag_dev, const float* phase_dev, int N, int T, int nfft, float* audio_dev,
               int device, void* stream);

/* The same two entries with the kernel family as an ARGUMENT (there is no process-wide state): RCED_AUDIO_X6 = the three-part bf16
 * kernels (fp32 quality on the bf16 matrix pipe, kernels_audio_x6.h: what rced_stft / rced_istft launch), RCED_AUDIO_F32 = the
 * fp32-MFMA kernels (the in-build comparator; both pass the same reference-pinned tests).  Any other value: RCED_ERR_ARG. */
#define RCED_AUDIO_F32 0
#define RCED_AUDIO_X6 1
int rced_stft_ex(const float* pcm_dev, const int* lengths_dev, int N, int L, int T, float* mag_dev,
                 float* phase_dev, int device, void* stream, int kernels);
int rced_istft_ex(const float* mag_dev, const float* phase_dev, int N, int T, int nfft, float* audio_dev,
                  int device, void* stream, int kernels);

/* ---- training step: FullyCNNTrainer.creat_graph + train_step,
 * model_utils/trainer.py:156-192, over Model(is_training=True).  Layer-by-layer, correctness first. ---- */
typedef struct rced_trainer rced_trainer;

/* blob: the same TF-variable blob rced_create takes (initial values, incl. BN moving statistics);
 * batch_size: the CONFIGURED batch size the loss divides by (trainer.py:146-147), not the dynamic N. */
int rced_train_create(int variant, const float* blob, size_t n_floats, int batch_size, int device,
                      rced_trainer** out);
void rced_train_destroy(rced_trainer* t);

/* sess.run([loss, global_step, train_op]) (trainer.py:181-192) with the UPDATE_OPS: forward with batch
 * statistics, loss = sum((y - pred)^2) / batch_size, backward, tf.train.AdamOptimizer(lr) step in TF's
 * form (beta1 0.9, beta2 0.999, eps 1e-8), moving mean / variance update (momentum 0.99).
 * x_dev, y_dev: DEVICE [N, T, 129, 1] float32.  lr: the value fed to the learning-rate placeholder.
 * *loss_out (host) receives the batch loss.  Synchronises the stream. */
int rced_train_step(rced_trainer* t, const float* x_dev, const float* y_dev, int N, int T, float lr,
                    double* loss_out, void* stream);

/* FullyCNNTrainer.valid_step (trainer.py:245-250): sess.run(self.pred) on the TRAINING graph, i.e. the model
 * built with is_training=True -- BatchNorm normalises with the statistics of the batch it is given, and because
 * only pred is fetched nothing is updated (no UPDATE_OPS, no optimizer step).  x_dev, pred_dev: DEVICE
 * [N, T, 129, 1] float32.  Synchronises the stream. */
int rced_train_forward(rced_trainer* t, const float* x_dev, float* pred_dev, int N, int T, void* stream);

long long rced_train_global_step(rced_trainer* t);
/* Current variables / last gradients, in blob order (gradients of moving statistics are 0). */
int rced_train_get_variables(rced_trainer* t, float* blob_host, size_t n_floats);
int rced_train_get_gradients(rced_trainer* t, float* blob_host, size_t n_floats);
/* Optimizer state, to save / resume a run as the reference does (trainer.py:50-65: tf.train.Saver(tf.global_variables())
 * stores the Adam slots "<var>/Adam", "<var>/Adam_1" and global_step next to the model variables).  m / v: first and
 * second moments in variable-blob order (entries of non-trainable variables are unused). */
int rced_train_get_state(rced_trainer* t, float* m_blob_host, float* v_blob_host, size_t n_floats, long long* global_step);
int rced_train_set_state(rced_trainer* t, const float* m_blob_host, const float* v_blob_host, size_t n_floats,
                         long long global_step);

/* The single op with is_training=True (module.py:29 `training=is_training`): BatchNorm normalises with the mean and the
 * BIASED variance of this batch over N*T*F (eps 1e-3), then + skip_input, then ReLU.  gamma_beta: gamma[cout], beta[cout]
 * (device).  batch_mean_var_out: NULL, or device [2*cout] receiving (mean, biased variance) -- what TF's UPDATE_OPS fold
 * into moving_mean / moving_variance (momentum 0.99; the variance Bessel-corrected by the fused kernel); the op itself
 * updates nothing, as the TF op does not unless the UPDATE_OPS are run.  use_norm=False has no training form: use
 * rced_conv_bn_relu with bn = NULL.  Direct-convolution kernels; synchronises the stream. */
int rced_conv_bn_relu_train(const float* x, float* y, const float* kernel, const float* bias, const float* gamma_beta,
                            const float* skip_input, int use_act, int N, int T, int F, int cin, int cout, int kh, int kw,
                            float* batch_mean_var_out, int device, void* stream);

/* Average device time (ms) of the dominant kernel of the last rced_forward, measured with HIP
 * events on the launch stream when profiling is on ("profile" option = 1).  <0 if none. */
float rced_last_kernel_ms(rced_model* m);

/* HIP-event profiler ("profile" option = 1 arms it and clears old samples): total device time
 * and launch count of kernel kind 0 = generic layer, 1 = fused multi-layer kernel, 2 = final
 * 1x129 Toeplitz GEMM, over every rced_forward since it was armed.  Synchronises. */
int rced_profile_query(rced_model* m, int kind, float* total_ms, int* launches);

/* Thread-local description of the last error on this thread ("" if none). */
const char* rced_last_error(void);

/* Library / build identification, e.g. "rced-hip 0.1 gfx950". */
const char* rced_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RCED_H_ */
