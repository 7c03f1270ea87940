/* iss.h -- C ABI of the MI355X-native inaSpeechSegmenter hot path (libiss_hip.so).
 *
 * The reference (ina-foss/inaSpeechSegmenter) has NO native/FFI boundary: its hot
 * path is numpy + TensorFlow/Keras + onnxruntime called from Python.  The entry
 * points below are what a ctypes binding inside the reference's own modules would
 * call instead; each one cites the reference interface it replaces (paths are
 * relative to the reference repo root, inaSpeechSegmenter/<file>:<line>).
 * INTEGRATION.md shows the reference-side ctypes stub.
 *
 * Conventions
 *   - plain C types only; no torch / numpy types cross this boundary.
 *   - every function returns 0 on success, a negative ISS_E* code on failure;
 *     iss_last_error(ctx) (ctx may be NULL for creation errors) gives the text.
 *   - the caller owns every host buffer; the library owns all device memory,
 *     streams and events and frees them in iss_destroy().
 *   - one context per (device, host thread); a context is not thread-safe.
 *   - there is NO CPU fallback: device entry points fail with ISS_ENODEV when no
 *     gfx950 device is usable.  The host-only helpers (iss_viterbi_*) need no GPU.
 */
#ifndef ISS_H
#define ISS_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISS_OK        0
#define ISS_EINVAL   -1   /* bad argument / bad program                         */
#define ISS_ENODEV   -2   /* no usable HIP device                               */
#define ISS_EHIP     -3   /* a HIP runtime call failed (text in last_error)     */
#define ISS_ESTATE   -4   /* call order violated (e.g. features before signal)  */
#define ISS_ENOMEM   -5
#define ISS_ETIMEOUT -6   /* a collective did not complete (peer rank gone): communicator aborted */

typedef struct iss_ctx iss_ctx;

/* ------------------------------------------------------------------ context */
int         iss_create(int device_id, iss_ctx** out);
void        iss_destroy(iss_ctx* ctx);
const char* iss_last_error(const iss_ctx* ctx);
const char* iss_version(void);
/* Cap (bytes) on the activation workspace of the CNN engine; decides how many
 * 20 ms slots are pushed through the layer stack per pass.  Default 24 GiB.    */
int         iss_set_workspace_limit(iss_ctx* ctx, uint64_t bytes);
int         iss_synchronize(iss_ctx* ctx);

/* ---------------------------------------------------- SIDEKIT log-mel front end
 * replaces sidekit_mfcc.py:278-352 `mfcc(sig, get_mspec=True)` as called from
 * segmenter.py:58 (framing 400/160, per-frame pre-emphasis, log-energy, Hann,
 * rfft-512 in float64, 24-band mel, log).                                      */

/* Constant tables (built by the host mirror with the reference's formulas):
 * window  = numpy.hanning(400) float64         (sidekit_mfcc.py:223)
 * melbank = trfbank(16000,512,100,8000,0,24)[0], (24,257) float32 row-major
 *           (sidekit_mfcc.py:118-197,332).                                     */
int iss_sidekit_tables(iss_ctx* ctx, const double* window400, const float* melbank_24x257);

/* Hand a decoded 16 kHz mono signal to the device (host -> HBM copy).
 * pcm16: what `ffmpeg -acodec pcm_s16le` produces (io.py:61-68); converted on the
 * device as x/32768 exactly like libsndfile's float read (io.py:77).
 * f32  : what soundfile hands back for float WAVs (io.py:52).                  */
int iss_signal_pcm16(iss_ctx* ctx, const int16_t* pcm, int64_t n);
int iss_signal_f32(iss_ctx* ctx, const float* sig, int64_t n);
/* Same, but the samples are ALREADY in device memory (hipMalloc'ed pointer, e.g. a
 * torch tensor's data_ptr); no copy, the buffer must outlive the feature call.
 * ORDERING CONTRACT: the library works on its own non-blocking stream.  The call makes that
 * stream wait (hipStreamWaitEvent) for everything submitted so far to `producer_stream`
 * (a hipStream_t; NULL = the legacy default stream, which is what torch uses unless the
 * caller switched streams), so a tensor produced there -- async H2D copy, kernel output --
 * is complete before the front end reads it.  Work the caller submits to OTHER streams is
 * the caller's to synchronise.  The pointer must belong to the context's device
 * (checked with hipPointerGetAttributes; ISS_EINVAL otherwise).                         */
int iss_signal_pcm16_device(iss_ctx* ctx, const void* dev_pcm, int64_t n);
int iss_signal_pcm16_device_stream(iss_ctx* ctx, const void* dev_pcm, int64_t n, void* producer_stream);

/* Page-locked host memory (hipHostMalloc) for decode buffers and result arrays: copies
 * from / to it are truly asynchronous (pageable memory is staged by the runtime).        */
int iss_host_alloc(iss_ctx* ctx, size_t bytes, void** out);
int iss_host_free(iss_ctx* ctx, void* p);

/* Run the front end on the resident signal.  Results stay in HBM (mspec (T,24),
 * loge (T,)); *T_out = int((n-400)/160)+1 (sidekit_mfcc.py:254), 0 if n < 400. */
int iss_sidekit(iss_ctx* ctx, int32_t* T_out);
int iss_get_loge(iss_ctx* ctx, float* loge_out /* T */);
int iss_get_mspec(iss_ctx* ctx, float* mspec_out /* T*24 */);
/* Replace the resident mel spectrogram (segment_feats(mspec, ...) entry,
 * segmenter.py:250, and the <68-frame padding of segmenter.py:61-65).          */
int iss_set_mspec(iss_ctx* ctx, const float* mspec, int32_t T);

/* ------------------------------------------------------------ small-CNN engine
 * replaces segmenter.py:76-88 `_get_patches` + :156-163 gather + `nn.predict`.
 *
 * A network is a flat op program (ISS_OP_* rows of ISS_PROG_COLS int32) plus one
 * float32 parameter blob; inaspeechsegmenter_amd/keras_model.py compiles a Keras
 * `model_config` into it.  net_id in [0, ISS_MAX_NETS).                        */
#define ISS_MAX_NETS   8
#define ISS_PROG_COLS  32

enum {
    ISS_OP_CONV     = 1,  /* conv2d / dense as implicit GEMM (MFMA) + fused epilogue (+ fused pool) */
    ISS_OP_POOL     = 2,  /* max / average pooling, NHWC; a padded (PT / PL, 'same') average leaves the padding out of the mean */
    ISS_OP_SOFTMAX  = 3,  /* softmax over the channel axis                               */
    ISS_OP_STATPOOL = 4,  /* mean || std over the time axis (resnet.py:123-127)          */
    ISS_OP_ACT      = 5,  /* elementwise activation beyond the fused ones: ISS_C_ACT 4 elu(alpha), 5 leaky relu(alpha),
                             6 selu, 7 softplus, 8 relu clipped at alpha (ReLU(max_value)); alpha = the float whose bits are in
                             ISS_C_ACTPARAM; 9 keras.layers.ReLU in full: x > threshold ? min(x, max_value) : negative_slope *
                             (x - threshold), (negative_slope, max_value (+inf: none), threshold) in ISS_C_ACTPARAM, ..2, ..3;
                             IN may equal OUT, and is never ISS_BUF_INPUT */
    ISS_OP_ELT      = 6,  /* merge / data-movement rows of graph-shaped models (keras.layers.Add, Concatenate, Permute ...; what
                             `keras.models.load_model`, segmenter.py:129-131, accepts beyond a chain).  Plain one-thread-per-element
                             kernels: correct, not fast.  ISS_C_ACT = kind (ISS_ELT_*); IN is never ISS_BUF_INPUT.
                             kinds ADD .. AVG: OUT[i] = IN[i] (op) RES[i] over H * W * CIN floats per sample (HO, WO, COUT = H, W, CIN;
                                               OUT may be IN or RES); ISS_C_ORDER = 1: followed by relu (Add + ReLU of a residual block);
                             COPY:    OUT[p][PL + c] = IN[p][PT + c] for c < KH, p over the H * W pixels; IN holds CIN and OUT COUT
                                      channels per pixel (HO, WO = H, W); OUT != IN.  Concatenate = one COPY per input;
                             ZERO:    OUT[p][PL + c] = 0 for c < KH (channel padding behind a COPY);
                             PERMUTE: OUT = IN with its (H, W, CIN) axes permuted, output axis i = input axis perm[i],
                                      perm = (KH, KW, SH); (HO, WO, COUT) = the permuted shape; OUT != IN */
};
enum { ISS_ELT_COPY = 0, ISS_ELT_ADD = 1, ISS_ELT_SUB = 2, ISS_ELT_MUL = 3, ISS_ELT_MAX = 4, ISS_ELT_MIN = 5, ISS_ELT_AVG = 6,
       ISS_ELT_ZERO = 7, ISS_ELT_PERMUTE = 8 };
/* column meaning of a program row (unused columns = 0, absent offsets = -1) */
enum {
    ISS_C_OP = 0, ISS_C_IN, ISS_C_OUT, ISS_C_RES,      /* buffer ids; RES = residual add   */
    ISS_C_H, ISS_C_W, ISS_C_CIN, ISS_C_HO, ISS_C_WO, ISS_C_COUT,
    ISS_C_KH, ISS_C_KW, ISS_C_SH, ISS_C_SW, ISS_C_PT, ISS_C_PL,
    ISS_C_ACT,                                          /* CONV: 0 none 1 relu 2 sigmoid 3 tanh (fused); ISS_OP_ACT rows: 4..9 */
    ISS_C_WOFF, ISS_C_BOFF,                             /* blob offsets: W [Cout][roundup32(kh*kw*Cin)] (WOFF % 8 == 0), bias */
    ISS_C_PSOFF, ISS_C_PTOFF,                           /* post-activation scale / shift   */
    ISS_C_INMODE,                                       /* 0 NHWC buffer, 1 z-normed mspec patch,
                                                           2 window of the resident vbx features (iss_vbx_embed) */
    ISS_C_POOLKIND,                                     /* POOL: 0 max 1 avg               */
    ISS_C_ORDER,                                        /* STATPOOL out order: 0 = (c,h) torch flatten */
    ISS_C_FPOOLH, ISS_C_FPOOLW,                         /* CONV: fused non-overlapping pool window applied after
                                                           the epilogue (0/1 = none; FPOOLH*FPOOLW in {2,4};
                                                           kind in ISS_C_POOLKIND).  HO/WO stay the conv's own
                                                           output size; OUT holds (HO/FPOOLH, WO/FPOOLW, COUT) */
    ISS_C_DUALW, ISS_C_DUALB,                           /* CONV, optional (0 = absent, else 1 + blob offset): row r is a 1x1
                                                           stride-1 conv whose residual (RES == OUT) is the output of row
                                                           r - 1, a LINEAR 1x1 conv (any stride, no bias-free form) of the
                                                           same output shape -- a projection shortcut and the expansion it
                                                           is added to (resnet.py:60-75).  DUALW: the concatenated matrix
                                                           [COUT][CIN(r) + CIN(r-1)] = [W(r) | W(r-1)], DUALB: b(r) + b(r-1).
                                                           The library may then compute both rows as ONE GEMM over the two
                                                           inputs (row r - 1's output is never materialised); it falls back
                                                           to the two rows whenever it cannot.  Both CIN % 32 == 0. */
    ISS_C_ACTPARAM,                                     /* ISS_OP_ACT: the bits of the activation's float parameter (alpha of elu / leaky relu) */
    ISS_C_ACTPARAM2, ISS_C_ACTPARAM3,                   /* ISS_OP_ACT code 9: max_value, threshold */
};
#define ISS_BUF_INPUT  (-2)   /* IN: the network input (patch source or iss_cnn_forward input) */

/* nbuf activation buffers with buf_elems[i] floats per sample; the last op's OUT
 * buffer holds the (out_dim,) result per sample.                                */
int iss_cnn_load(iss_ctx* ctx, int net_id, const int32_t* prog, int32_t nrows,
                 const float* blob, int64_t blob_floats,
                 int32_t nbuf, const int64_t* buf_elems,
                 int32_t in_h, int32_t in_w, int32_t in_c, int32_t out_dim);

/* Class probabilities of n 20 ms slots.  win_row[i] = first mspec row of the 68-frame
 * window feeding slot i (the host applies the 17-left/16(+1)-right edge replication of
 * segmenter.py:83-84 when it builds this list); nmel columns are used (21 or 24,
 * segmenter.py:146-147).  Per window: z-normalisation with population std
 * (segmenter.py:82) and finite_out[i] = all(isfinite(normalised patch)) (:86); windows
 * that are not finite get probs 0.5 (segmenter.py:175).                         */
int iss_cnn_probs(iss_ctx* ctx, int net_id, const int32_t* win_row, int32_t n,
                  float* probs_out /* n*out_dim */, uint8_t* finite_out /* n */);

/* Asynchronous form: enqueues the same work on the context's stream and returns; win_row
 * is consumed before the call returns, probs_out / finite_out (page-locked memory from
 * iss_host_alloc if the copy is to overlap host work) are valid after iss_wait(ctx, ticket)
 * (or any other call that synchronises the context).  The resident mel spectrogram must not
 * be replaced before iss_wait.  Lets the host run the Viterbi of one network
 * (segmenter.py:176) while the device already evaluates the next one.                  */
int iss_cnn_probs_async(iss_ctx* ctx, int net_id, const int32_t* win_row, int32_t n,
                        float* probs_out /* n*out_dim */, uint8_t* finite_out /* n */, int64_t* ticket_out);
/* Block until the work of `ticket` (and everything enqueued before it) is complete; ticket < 0 = the whole stream. */
int iss_wait(iss_ctx* ctx, int64_t ticket);

/* Generic batched forward on caller-supplied host input (n, in_h, in_w, in_c) f32 NHWC:
 * replaces vbx_segmenter.py:262-266 `OnnxBackendExtractor.get_embedding` (ResNet-101
 * of resnet.py:78-135, one launch sequence for many windows instead of batch 1).  */
int iss_cnn_forward(iss_ctx* ctx, int net_id, const float* x, int32_t n, float* out /* n*out_dim */);

/* Arithmetic mode of the conv/dense GEMMs.  gfx950 has no xf32 / TF32 and its f32-input MFMA runs at 1/16 of the 16-bit rate, so
 * both operands are split x = hi + lo into 16-bit halves and every k-step issues three MFMAs (lo.hi + hi.lo + hi.hi, f32 accumulate):
 *   ISS_PREC_F16X3   (default since round 6) fp16 halves, 11 + 11 mantissa bits, v_mfma_f32_32x32x16_f16, in the kernels that carry the
 *                    segmenter nets' arithmetic (conv_x3_wq_kernel, conv_x3_wq3h_kernel, conv_x3_pw_kernel); exact f32 for their
 *                    small trailing layers; bf16 halves in every kernel without an fp16 instantiation.  Needs parameters and
 *                    activations inside fp16's range (|x| < 65504): parameters are checked at load (a network with a larger one
 *                    runs ISS_PREC_BF16X3), activations by the precision guard's probe.  max |d log p| against exact f32 on the
 *                    stand-ins: 4.7e-5 (profiles/r06_f16_ab.txt), the speed of ISS_PREC_BF16X3 within 1 %;
 *   ISS_PREC_BF16X3  bf16 halves, 8 + 8 mantissa bits, v_mfma_f32_32x32x16_bf16, in every GEMM kernel: operand error 2^-16 relative,
 *                    max |d log p| 2.9e-4 on the same data (the default of rounds 1-5);
 *   ISS_PREC_F32     v_mfma_f32_32x32x2_f32, bit-wise an fmaf chain (reference-grade, ~3 x slower).                          */
#define ISS_PREC_BF16X3 0
#define ISS_PREC_F32    1
#define ISS_PREC_F16X3  2
int iss_set_precision(iss_ctx* ctx, int mode);

/* Precision guard for weights nobody has measured (north star: "frame logits within 1e-3 fp32"; segmenter.py:163,176 --
 * the reference's emissions are log(predict(...)) in f32).  How far inside that bound a split mode sits depends on the activation
 * ranges of the weights actually loaded (tests/precision_emulation.py, profiles/r06_precision_emulation.txt).  So the FIRST
 * iss_cnn_probs / iss_cnn_probs_async call of a patch network in a split mode first runs up to 256 of the call's own windows
 * (four runs of consecutive slots spread over the list) in that mode and in exact f32, records max |log p_split - log p_f32| over
 * every class of every finite window, and -- when that exceeds `threshold` (default 5e-4, half the bound; a NaN, i.e. an activation
 * beyond fp16's range, always does) -- switches THIS network for the rest of its life: to the other split mode if that one passes
 * the same probe (the same speed), else to ISS_PREC_F32.
 * iss_set_precision_guard: threshold <= 0 disables the probe (networks loaded later are not probed; already decided ones keep their mode).
 * iss_cnn_precision_info: mode in use for the network (ISS_PREC_*), the probe's figure for the mode that was asked for (-1 if not
 * probed), the windows it compared, ISS_GUARD_* state, and the figure of the mode in use (0 for exact f32).
 * iss_cnn_set_net_precision: caller's override for one network (-1 = follow the context again); marks it decided. */
#define ISS_GUARD_PENDING   0   /* not probed yet                                              */
#define ISS_GUARD_PASSED    1   /* probed: within the threshold, mode kept                     */
#define ISS_GUARD_ESCALATED 2   /* probed: above the threshold, the network runs another mode  */
#define ISS_GUARD_FIXED     3   /* mode set by the caller (iss_cnn_set_net_precision) or the context is in exact-f32 mode anyway */
int iss_set_precision_guard(iss_ctx* ctx, float threshold);
int iss_cnn_precision_info(iss_ctx* ctx, int id, int32_t* mode, float* max_dlogp, int32_t* slots, int32_t* state, float* dlogp_in_use);
int iss_cnn_set_net_precision(iss_ctx* ctx, int id, int mode);

/* Kernel-selection switches for same-box A/B measurements and for the tests that compare two device paths with each
 * other (e.g. the shared first layer against the per-window one).  0 (default) = production selection.  The library never
 * reads the environment for these: inaspeechsegmenter_amd/_native.py maps the ISS_DIAG environment variable (a
 * comma-separated list of the names below, lower case, without the prefix) onto this call when a context is created.  */
#define ISS_DIAG_NO_SHARED_FIRST 0x001u  /* per-window first layer instead of the shared one (segmenter.py:82 per window)   */
#define ISS_DIAG_NO_FLROWS       0x002u  /* first_layer_raw_kernel instead of first_layer_rows_kernel                        */
#define ISS_DIAG_NO_WS           0x004u  /* no weight-stationary footprint kernel (conv_x3_fp_kernel everywhere)             */
#define ISS_DIAG_NO_WS3          0x008u  /* no NH = 2 weight-stationary kernel for the 3x3 layers                            */
#define ISS_DIAG_NO_DIRECT1      0x010u  /* no direct f32 kernel for one-channel 3x3 first layers                            */
#define ISS_DIAG_NO_NH2          0x020u  /* conv_x3_fp_kernel: 64 output channels per workgroup                              */
#define ISS_DIAG_NO_TR           0x040u  /* conv_x3_fp_kernel: row-major epilogue                                            */
#define ISS_DIAG_NO_PW           0x080u  /* 1x1 layers on the generic gather kernel                                          */
#define ISS_DIAG_NO_PWS          0x100u  /* 1x1 layers on the round-2 pointwise kernel                                       */
#define ISS_DIAG_NO_PWS2         0x200u  /* 1x1 layers: 64-column tiles only                                                 */
#define ISS_DIAG_NO_WQ           0x400u  /* conv_x3_ws_kernel (two waves per SIMD) instead of conv_x3_wq_kernel for the fused 5x3 layer */
#define ISS_DIAG_NO_DUAL         0x800u  /* projection shortcut + expansion as two launches (ISS_C_DUALW ignored)            */
#define ISS_DIAG_NO_CHAIN        0x1000u /* identity-residual expansion and the next block's reduction as two launches       */
#define ISS_DIAG_NO_RING         0x2000u /* no ring form of the weight-stationary kernel (second convs with > 16 taps: gather kernel) */
#define ISS_DIAG_NO_FSAME        0x4000u /* a zero-padded ('same') first layer is run per window, not shared between windows  */
#define ISS_DIAG_NO_F32WS        0x8000u /* exact-f32 mode: conv_igemm_kernel everywhere (no F32 form of the weight-stationary kernel) */
#define ISS_DIAG_NO_NCB1         0x10000u /* layers with <= 32 output channels on the 64-column forms (no NCB = 1 form)                 */
#define ISS_DIAG_NO_WSU3         0x20000u /* unpadded 3x3 layers with 64 / 96 output channels on conv_x3_fp_kernel                      */
#define ISS_DIAG_NO_GFUSED       0x40000u /* the generic gather kernel never reads the shared first-layer rows (per-window first layer)  */
#define ISS_DIAG_NO_HL           0x80000u /* f32 NHWC activations between the footprint kernels (no CHL layout: conv_x3_wq3_kernel instead of conv_x3_wq3h_kernel) */
#define ISS_DIAG_ALL             0xfffffu
int iss_set_diag(iss_ctx* ctx, uint32_t flags);

/* FLOPs (2*MAC of the conv/dense outputs actually computed) per sample of a loaded network. */
int iss_cnn_flops(iss_ctx* ctx, int net_id, double* flops_per_sample);

/* ------------------------------------------------ VBx 64-band fbank front end
 * replaces vbx_segmenter.py:72-89 `get_features` (features_vbx.py:62-149).
 * window = povey_window(400) f64; melbank = mel_fbank_mx(...) (257,64) f64 row-major.
 * sig_i32 = (signal * 2**15).astype(int) (vbx_segmenter.py:85); dither_u = the
 * np.random.seed(3) uniform stream of n doubles (features_vbx.py:127-128), generated
 * on the host because it is an MT19937 stream.  out = (T,64) f32, T = n/160 style
 * count *T_out = (n + 320 - 400)/160 + 1.                                        */
int iss_vbx_tables(iss_ctx* ctx, const double* window400, const double* melbank_257x64);
int iss_vbx_features(iss_ctx* ctx, const int32_t* sig_i32, const double* dither_u, int64_t n,
                     float* fea_out /* T*64, may be NULL to keep on device */, int32_t* T_out);
/* Same for PCM16 sources (what io.py's ffmpeg hop yields: (signal * 2**15).astype(int) is then the PCM itself) with a
 * dither stream cached on the device: np.random.seed(3) gives every file the same stream, so upload its longest prefix
 * once (iss_vbx_set_dither) and send 2 bytes per sample afterwards.                                                    */
int iss_vbx_set_dither(iss_ctx* ctx, const double* dither_u, int64_t n);
int iss_vbx_features_pcm16(iss_ctx* ctx, const int16_t* pcm, int64_t n,
                           float* fea_out /* T*64 or NULL */, int32_t* T_out);
/* x-vectors of n windows [starts[i], starts[i] + frames) of the RESIDENT features (frames = the loaded network's input
 * width; its first conv must be a window-mode conv, ISS_C_INMODE 2): replaces the window loop of
 * vbx_segmenter.py:222-231 + get_embedding (:262-266) without copying windows through the host.                        */
int iss_vbx_embed(iss_ctx* ctx, int net_id, const int32_t* starts, int32_t n, float* out /* n*out_dim */);

/* ------------------------------------------- multi-GPU: the single exchange step
 * Long archives shard file-parallel over the GPUs of one node (one process per GPU; files
 * are independent units, segmenter.py:314-327 / scripts/ina_speech_segmenter_pyro_server.py:
 * 34-68).  After local processing every rank holds a table of int32 rows
 * (file_id, label_id, start_slot, stop_slot); ONE ncclAllGather (RCCL over xGMI) of
 * fixed-capacity buffers with a header row (n_rows, capacity, rank, 0) leaves every
 * rank's rows on every rank.  librccl is dlopen'ed on first use: single-GPU users do not
 * need it.  Rendezvous: rank 0 calls iss_comm_unique_id and hands the 128 bytes to the
 * other ranks by any out-of-band means (inaspeechsegmenter_amd/sharding.py uses a TCP
 * socket on MASTER_ADDR); then every rank calls iss_comm_init.                        */
#define ISS_COMM_ID_BYTES 128
int iss_comm_unique_id(iss_ctx* ctx, uint8_t id_out[ISS_COMM_ID_BYTES]);
int iss_comm_init(iss_ctx* ctx, const uint8_t id[ISS_COMM_ID_BYTES], int32_t rank, int32_t world);
int iss_comm_destroy(iss_ctx* ctx);
/* local_rows: (n_local,4) int32.  all_rows: (world*capacity,4) int32, rank r's rows start at
 * r*capacity; counts[r] = rows rank r HAS (may exceed capacity: then only the first
 * `capacity` arrived and the caller repeats the call with capacity >= max(counts)).
 * Failure handling (the reference's Pyro workers fail independently, scripts/ina_speech_segmenter_pyro_client.py:64-74;
 * a collective cannot): a rank whose local work failed still takes part with n_local = -1, every rank then sees
 * counts[r] == -1 and raises; a rank that is GONE makes the others wait ISS_COMM_TIMEOUT_S seconds (environment,
 * default 1800), after which the communicator is aborted (ncclCommAbort) and the call returns ISS_ETIMEOUT.          */
int iss_allgather_segments(iss_ctx* ctx, const int32_t* local_rows, int32_t n_local, int32_t capacity,
                           int32_t* all_rows, int32_t* counts /* world */);
/* max over ranks of a double (bench timing) and a barrier, on the same communicator */
int iss_comm_allreduce_max(iss_ctx* ctx, double* value);
/* what RCCL itself reports for the communicator: ncclCommCount / ncclCommUserRank / ncclGetVersion and the path of
 * the librccl that was bound (audit trail of the multi-GPU bench line) */
int iss_comm_info(iss_ctx* ctx, int32_t* world, int32_t* rank, int32_t* version, char* lib_path, int32_t lib_path_len);

/* ------------------------------------------------------------ profiling hooks */
/* Accumulated device time (ms, hipEvent-timed on the context's stream), launch count and algorithmic flops of the
 * kernel classes since the last reset.  kind: 0 = all conv/dense implicit-GEMM kernels, 1 = sidekit / vbx fbank front
 * end, 2 = everything else; 3.. = the GEMM kernels one by one (each launch counted in 0 AND in its own class).       */
#define ISS_PROF_GEMM      0
#define ISS_PROF_FRONTEND  1
#define ISS_PROF_OTHER     2
#define ISS_PROF_WS        3   /* conv_x3_ws_kernel   (weight-stationary footprint kernel) */
#define ISS_PROF_FP        4   /* conv_x3_fp_kernel   (streaming-weights footprint kernel) */
#define ISS_PROF_PW        5   /* conv_x3_pw_kernel   (1x1 / dense persistent GEMM) */
#define ISS_PROF_GATHER    6   /* conv_x3_kernel      (generic gather GEMM) */
#define ISS_PROF_PATCH1    7   /* conv1_patch_x3_kernel (per-window first layer) */
#define ISS_PROF_F32       8   /* conv_igemm_kernel   (exact-f32 MFMA mode) */
#define ISS_PROF_KINDS     9
int iss_prof_enable(iss_ctx* ctx, int on);
int iss_prof_get(iss_ctx* ctx, int kind, double* ms, int64_t* launches, double* flops);
int iss_prof_reset(iss_ctx* ctx);
/* Per-layer view of the same HIP-event brackets: time and launch count of op-program row `row` (the conv / dense row of
 * the loaded network that a launch executed; rows of different networks share the index space) since the last reset. */
#define ISS_PROF_ROWS      512
int iss_prof_get_row(iss_ctx* ctx, int row, double* ms, int64_t* launches);
/* Per kernel INSTANTIATION: entry `index` (0, 1, ... until ISS_EINVAL) of the list of distinct kernel instantiations
 * launched since the last reset -- name with its template arguments spelled out (e.g.
 * "conv_x3_ws_kernel<5,3,false,false,true,1,1>"), accumulated HIP-event time, launches and algorithmic flops.  This is what
 * bench.py's roofline.dominant is computed from.                                                                        */
int iss_prof_get_instance(iss_ctx* ctx, int index, char* name_out, int32_t name_len, double* ms, int64_t* launches, double* flops);

/* ------------------------------------------------------------------ host only
 * Viterbi smoothing, replaces pyannote_viterbi.py:118-224 `viterbi_decoding` on the
 * unconstrained path the segmenter uses (segmenter.py:72-73,176): float64 scores,
 * uniform initial log(1/K), argmax takes the FIRST maximum, back-tracking :217-220.
 * emission (T,K) row-major; f32 variant promotes each value to double exactly like
 * numpy does when `np.log(r)` (float32) is added to float64 (segmenter.py:176).
 * transition (K,K) row-major, T[i][j] = i -> j.  K <= 16.                        */
int iss_viterbi_f64(const double* emission, int64_t T, int32_t K, const double* transition,
                    int32_t* states_out);
int iss_viterbi_f32(const float* emission, int64_t T, int32_t K, const double* transition,
                    int32_t* states_out);
/* nseg consecutive segments of one (sum(seg_len), K) emission array, each smoothed on its own like iss_viterbi_f32
 * (the per-segment loop of segmenter.py:168-178 in one call).                                                          */
int iss_viterbi_segments_f32(const float* emission, const int64_t* seg_len, int64_t nseg, int32_t K,
                             const double* transition, int32_t* states_out);
/* `_energy_activity` behind its threshold (segmenter.py:69-73): raw activity loge[t] > threshold (float64 compare),
 * pred2logemission (viterbi_utils.py:29-34; the caller passes numpy's log(eps) and log(1 - eps)), two-state Viterbi with the
 * (2,2) transition matrix of log_trans_exp (viterbi_utils.py:36-42).  states_out[t] in {0, 1}.                         */
int iss_energy_viterbi(const float* loge, int64_t T, double threshold, double log_eps, double log_1m_eps,
                       const double* transition, int32_t* states_out);

#ifdef __cplusplus
}
#endif
#endif /* ISS_H */
