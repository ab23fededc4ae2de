/*
 * diffsci_hip.h -- C ABI of libdiffsci_hip.so: the MI355X (gfx950) kernels behind the
 * Karras-EDM sampling path of DiffSci.
 *
 * The reference (Lacadame/DiffSci) is pure Python on PyTorch; it has no FFI of its own.
 * The boundary it exposes for this path is the Python object protocol
 * (KarrasModule.sample / Scheduler.propagate / Integrator.step / model(x, c_noise, y)),
 * which diffsci_amd mirrors in Python.  Below that protocol every per-step tensor
 * operation the reference issues as ATen calls is one of the entry points declared
 * here; each cites the reference code it replaces (paths relative to the reference
 * root).
 *
 * Conventions
 *   - plain C, no torch types: raw device pointers, sizes, a hipStream_t passed as void*;
 *   - every function returns 0 on success, a negative ds_status otherwise; the text of
 *     the last failure on the calling thread is available from ds_last_error();
 *   - all pointers are caller-owned device memory (fp32, contiguous NCHW unless stated);
 *     nothing is allocated or freed inside, nothing synchronises with the host, so every
 *     launch may be captured into a hipGraph (ds_graph_*);
 *   - launches are asynchronous on the given stream; distinct streams may be driven from
 *     distinct host threads;
 *   - argument shapes are validated on the host before any launch (DS_ERR_SHAPE);
 *   - the elementwise stepper entry points take any element count and any 4-byte-aligned
 *     pointers (16-byte-aligned operands run the float4 path, others a scalar path with
 *     identical arithmetic): the reference accepts any tensor shape.
 *
 * Environment (read once per process; measurement switches, results are identical up to fp32 rounding of the accumulation
 * order): DIFFSCI_HIP_LIB = path of another build of this library (A/B runs, diffsci_amd/_native.py); DS_CONV_SHAPE=32 =
 * the v_mfma_f32_32x32x16_f16 form of ds_conv2d_h3 / ds_conv2d_h3_up everywhere (default: 16x16x32 wherever the layer has
 * an even number of 16-channel chunks); DS_CONV_WAVES=4|8 = waves per workgroup of the 32x32x16 form (default: 8 with the
 * fused loader); DS_CONV_WAVES16=8 = eight waves for the 16x16x32 form with the fused loader (default 4); DS_ATTN_T =
 * rescaling threshold of ds_attention_h3's online softmax (default 8); DS_CONV_PC = 0|1|2|3 = which launches of ds_conv2d_h3 take the
 * persistent producer / consumer form (ds_conv3p.hip: 0 none, 1 fused-loader launches with one channel tile, 2 (default) also those
 * with several, 3 raw-input launches too; bit-identical results), DS_CONV_PC_MIN = fewest items per workgroup for it (default 1),
 * DS_CONV_PC_IMG=0 = ds_conv2d_h3_img stays on the one-shot kernel, DS_CONV_PC_SKEW = mask of the persistent kernel's start-up stagger
 * (default 0), DS_CONV_PC_PRIO = s_setprio level of its producer waves (default 0), DS_CONV_PC_WAVES=8 = eight producer waves (the
 * one-pixel staging plan only); DS_CONV_VEC=0 = one-pixel staging items instead of the 16-byte patch loads (both kernels),
 * DS_CONV_TWO=0|1|2 / DS_CONV_TWO_MIN / DS_CONV_TWO_EARLY=0 = the two-channel-tile one-shot kernel: off | two tiles | every even
 * count; its smallest grid; waves 0-3 staging after the step's matrix instructions like waves 4-7; DS_DIRECT_VEC=0 = ds_conv2d_direct's
 * general kernel on whole 64-column tiles too.
 */
#ifndef DIFFSCI_HIP_H
#define DIFFSCI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ds_status {
  DS_OK = 0,
  DS_ERR_SHAPE = -1,     /* an argument violates a documented shape/alignment rule */
  DS_ERR_NULL = -2,      /* required pointer is NULL */
  DS_ERR_HIP = -3,       /* a HIP runtime call failed (see ds_last_error) */
  DS_ERR_UNSUPPORTED = -4
} ds_status;

/* Library / device introspection. */
int ds_version(void);                       /* ABI version, currently 4 (DS_ABI_VERSION in ds_api.hip, ABI_VERSION in diffsci_amd/_native.py) */
const char* ds_last_error(void);            /* thread-local, never NULL */
int ds_device_info(int* cu_count, int* lds_bytes_per_cu, char* arch_name, int arch_name_len);

/* ------------------------------------------------------------------------------------
 * Karras stepper.  Per-evaluation scalars come from a host-built table: in one sampling
 * run every sample of the batch sits at the same noise level (schedulers.py:254), so
 * sigma, dt and the preconditioner values are per-step constants computed once on the
 * host with the reference's own fp32 torch-CPU operation sequence.
 * ---------------------------------------------------------------------------------- */
typedef struct ds_eval_coef {
  float c_out;      /* preconditioners.py:41-44  output_scaling(sigma)                    */
  float c_skip;     /* preconditioners.py:35-39  skip_scaling(sigma)                      */
  float sigma_sq;   /* karrasmodule.py:733       sigma**2                                 */
  float neg_mult;   /* schedulers.py:266-268     -(sigma * sigma')                        */
  float neg_lang;   /* schedulers.py:269-274     -langevin_factor(t); 0 when deterministic */
  float guidance;   /* karrasmodule.py:709-713   g of (1-g)*F_u + g*F_c; unused if fu NULL */
  float one_minus_guidance; /* (1-g) rounded to fp32 from the host's double, as torch does     */
  int   input_kind; /* what `f` holds: 0 network output F (apply the preconditioner),
                       1 score (generic score_fn path of Scheduler.rhs), 2 drift (already rhs) */
  int   stochastic; /* 1: add neg_lang*score to the drift                                 */
  /* Non-constant scaling s(t) (VP; Scheduler.rhs, schedulers.py:275-293).  scaled = 0: the three fields are ignored and the
   * arithmetic is the constant-scaling branch's, bit for bit.  scaled = 1:
   *   x~ = x / scale;  score = score(x~)  [D = c_out*F + c_skip*x~, F = net(c_in*x~, c_noise)];
   *   d = scale_mult*x + neg_mult*score [+ neg_lang*score]   with neg_mult = -(s sigma' sigma), neg_lang = -(lambda*1/s)  */
  int   scaled;
  float scale;      /* s(t)                                                                */
  float scale_mult; /* s'(t) / s(t)                                                        */
  float next_scale; /* s at the evaluation the emitted network input (xin_out) feeds: xin_out = c_in_next * (x_out / next_scale)
                       when != 1 and != 0 (the reference calls score_fn(x / s, sigma), schedulers.py:287)                     */
  int   xin_copies; /* 2: xin_out holds TWO copies back to back ([2n] floats): classifier-free guidance evaluates the
                       conditional and the unconditional branch as one evaluation of batch 2B on the same input
                       (karrasmodule.py:706-713); 0 / 1: one copy                                                         */
  uint32_t* nonfinite; /* device word or NULL.  ds_karras_euler (k) / ds_karras_heun (k2) OR 1 into it when a value of x_out is
                       inf or NaN: the run-level result check of the range guard (a host-side isfinite(out).all() in round 2)
                       carried by the run's last step kernel; one atomic per wave that saw one, none on a finite run           */
} ds_eval_coef;

enum { DS_IN_NETWORK = 0, DS_IN_SCORE = 1, DS_IN_DRIFT = 2,
       DS_IN_FLOW = 3 /* f is a flow field (SIModule, flowfield.py:441-458): d = neg_mult*(blend(f, fu)/sigma_sq) */ };

/* out = s * x.  karrasmodule.py:881 (x * maximum_scale) and :702 (c_in * x). */
int ds_karras_scale(float* out, const float* x, float s, size_t n, void* stream);

/* d = drift(x, f) written out (see ds_karras_euler for the formula): Scheduler.rhs
 * (schedulers.py:247-274) composed with KarrasModule.get_score (karrasmodule.py:721-733) when
 * k->input_kind == 0, or Scheduler.rhs alone on a score tensor when input_kind == 1. */
int ds_karras_drift(float* d_out, const float* x, const float* f, const float* fu,
                    const ds_eval_coef* k, size_t n, void* stream);

/* Score (D - x)/sigma^2 from a network output: karrasmodule.py:717-733 (uniform sigma). */
int ds_karras_score(float* s_out, const float* x, const float* f, const float* fu,
                    const ds_eval_coef* k, size_t n, void* stream);

/* Standard-normal noise generated inside the kernels that consume it (SURVEY 8a6 option (i); the reference
 * draws torch.randn_like(x) per step: integrators.py:66-69,103-104).  Philox4x32-10, key = state[0] (seed),
 * counter = state[1] + philox_offset + e/4 for element e, whose four 32-bit outputs give, by Box-Muller, the
 * normals of elements 4*(e/4) .. 4*(e/4)+3.  `philox_state` is a DEVICE pointer to two uint64 {seed, base
 * offset}: a captured graph bakes the per-step `philox_offset` and re-reads the state on every replay, so the
 * host re-seeds a replay by rewriting 16 bytes.  The draw does not depend on launch geometry or alignment:
 * the same (seed, offsets) reproduce it bit for bit.  ds_philox_normal writes the stream out (tests, and
 * callers that want the eps a run used). */
int ds_philox_normal(float* out, const uint64_t* philox_state, uint64_t philox_offset, size_t n, void* stream);

/* Euler move from one evaluation:  d = drift(x, f);  x_out = x + dt*d  [+ (noise_coef*eps)*sqrt_abs_dt]
 * and, when xin_out != NULL, xin_out = c_in_next * x_out (the next evaluation's network input).
 *   drift: D = c_out*F + c_skip*x; score = (D - x)/sigma_sq; d = neg_mult*score [+ neg_lang*score]
 *   F = f, or (1-g)*fu + g*f when fu != NULL.
 * Replaces EulerIntegrator.step (integrators.py:29-35), the predictor half of
 * HeunIntegrator.step (integrators.py:44-47), EulerMaruyamaIntegrator.step (integrators.py:66-69:
 * eps != NULL injects the noise, philox_state != NULL generates it in the kernel, both NULL: none)
 * together with Scheduler.rhs (schedulers.py:247-274) and
 * KarrasModule.get_denoiser/get_score (karrasmodule.py:690-733).
 * x_out may be NULL (only xin_out wanted) ; x_out may alias x. */
int ds_karras_euler(float* x_out, float* xin_out, const float* x, const float* f, const float* fu,
                    const ds_eval_coef* k, float dt, float c_in_next,
                    const float* eps, const uint64_t* philox_state, uint64_t philox_offset,
                    float noise_coef, float sqrt_abs_dt, size_t n, void* stream);

/* Heun corrector:  x_out = x + (0.5*(d1 + d2))*dt, with d1 recomputed from (x, f1) and d2 from
 * (x_e = x + dt*d1, f2) -- x_e is recomputed, never read.  integrators.py:44-53.
 * When k2->input_kind != 0, f2 is the score / drift at x_e and x_e is not needed.
 * xin_out (optional) = c_in_next * x_out.  x_out may alias x. */
int ds_karras_heun(float* x_out, float* xin_out, const float* x,
                   const float* f1, const float* f1u, const ds_eval_coef* k1,
                   const float* f2, const float* f2u, const ds_eval_coef* k2,
                   float dt, float c_in_next, size_t n, void* stream);

/* Noise injection x_hat = ratio*x + coef*eps (ratio == 1: x + coef*eps); xin_out (optional) = c_in * x_hat, or
 * c_in * (x_hat / scale) when scale != 1.  KarrasIntegrator.step sigma-churn (integrators.py:98-105: ratio = s(t_hat)/s(t),
 * coef = std*s_noise, scale = s(t_hat)) and Scheduler.renoise (schedulers.py:166-176).  Exactly one of eps (injected, [n]) and
 * philox_state (generated in the kernel: 12 B per element instead of 16, no eps buffer) is given. */
int ds_karras_churn(float* xhat_out, float* xin_out, const float* x, const float* eps,
                    const uint64_t* philox_state, uint64_t philox_offset,
                    float coef, float c_in, float ratio, float scale, int xin_copies /* as ds_eval_coef.xin_copies */,
                    size_t n, void* stream);

/* Denoiser with per-sample coefficients (sigma differs across the batch):
 * out = c_out[b]*F + c_skip[b]*x, F as above.  karrasmodule.py:717-718.  Coefficient arrays are
 * device pointers of length B; n_per_sample = C*H*W. */
int ds_karras_denoiser(float* out, const float* x, const float* f, const float* fu, float guidance,
                       float one_minus_guidance,
                       const float* c_out, const float* c_skip, int B, size_t n_per_sample, void* stream);

/* ------------------------------------------------------------------------------------
 * Score network (PUNetG) layers.
 * ---------------------------------------------------------------------------------- */

/* Per-(sample, channel) normalisation over H*W fused with SiLU:
 *   kind 0: GroupNorm(num_groups=C) -- (x-mean)/sqrt(var_biased+eps)*w[c]+b[c]  (commonlayers.py:766-770,824)
 *   kind 1: GroupRMSNorm(C, C)      -- x/sqrt(mean(x^2)+eps)*w[c]+b[c]          (commonlayers.py:372-384,829)
 *   kind 2: no normalisation (first/second_resblock_norm other than the named ones -> Identity, commonlayers.py:891-899)
 *   kind 3: GroupPixNorm(C, C)      -- x/sqrt(x^2+eps)*w[c]+b[c]                (commonlayers.py:387-440)
 * followed by x*sigmoid(x).  x, out: [B, C, HW]. out may alias x.  w, b may be NULL (affine=False). */
int ds_inorm_silu(float* out, const float* x, const float* w, const float* b,
                  int B, int C, int HW, float eps, int kind, void* stream);

/* Weight repacking for ds_conv2d: torch layout [Cout, Cin, ks, ks] -> the MFMA A-operand stream
 * [ceil(Cout/64)][ceil(Cin/8)][ks*ks][8][64], zero padded.  ds_conv2d_packed_floats gives the
 * destination size in floats. */
size_t ds_conv2d_packed_floats(int Cout, int Cin, int ks);
int ds_conv2d_pack_weights(float* packed, const float* w, int Cout, int Cin, int ks, void* stream);

enum { DS_LOAD_PLAIN = 0, DS_LOAD_MAXPOOL2 = 1, DS_LOAD_UPSAMPLE2 = 2, DS_LOAD_AVGPOOL2 = 3 /* ds_conv1x1_h3 only */,
       DS_RES1_UPSAMPLED = 32 /* OR-ed into load_mode of ds_conv2d_h3: res1 is [B, Cout, H/2, W/2] and is added
                                 nearest-upsampled -- ADM's convresidual(upsample(x)) = upsample(convresidual(x)), adm.py:345-349 */,
       DS_PAD_CIRCULAR = 16 /* OR-ed into load_mode of ds_conv2d_h3: periodic instead of zero padding in H and W
                               (CircularConv2d, commonlayers.py:918-971; applied to the pooled / upsampled image) */ };
/* OR-ed into load_mode of ds_conv2d_h3: the 3x3 window is centred at (y + oy, x + ox) instead of (y, x), -8 <= oy, ox <= 7
 * (in the coordinates of the pooled / upsampled image; zero padding).  A k x k kernel with k = 5, 7, ... (kernel_size /
 * in_out_kernel_size / transition_kernel_size of PUNetGConfig, punetg_config.py:19-25) is the sum of ceil(k/3)^2 such
 * 3x3 convolutions over zero-padded blocks of its taps, accumulated through res1 = out. */
#define DS_TAP_OFFSET(oy, ox) ((((oy) & 15) << 8) | (((ox) & 15) << 12))

/* "same"-padded (zero) ks x ks convolution, ks in {1,3}, fp32 MFMA implicit GEMM.
 *   out[b,co,y,x] = sum w[co,ci,ky,kx]*src(b,ci,y+ky-ks/2,x+kx-ks/2) + bias[co]
 *                   + shift[b*shift_stride + co] + res1[b,co,y,x] + res2[b,co,y,x]
 * (each optional term skipped when its pointer is NULL; added in that order).
 *   load_mode PLAIN:     src = in,                      in is [B,Cin,H,W]
 *             MAXPOOL2:  src = maxpool2x2(in),          in is [B,Cin,2H,2W]     (DownSampler, commonlayers.py:81)
 *             UPSAMPLE2: src = nearest_upsample2x(in),  in is [B,Cin,H/2,W/2]   (UpSampler, commonlayers.py:145)
 * out is [B,Cout,H,W].  Replaces conv1 + time shift (commonlayers.py:824-828), conv2 + residual
 * (commonlayers.py:829-833), convin/convout (punetg.py:395,415), the skip add (punetg.py:374),
 * x + xa (punetg.py:385), and the MultiheadAttention in/out projections (ks = 1). */
int ds_conv2d(float* out, const float* in, const float* w_packed, const float* bias,
              const float* shift, int shift_stride, const float* res1, const float* res2,
              int B, int Cin, int Cout, int H, int W, int ks, int load_mode, void* stream);

/* The same 3x3 convolution computed on the bf16 matrix cores with fp32 accuracy ("bf16x6"):
 * each fp32 operand is split exactly into three bf16 pieces and the six leading piece products
 * are accumulated in fp32 (v_mfma_f32_32x32x16_bf16); dropped terms are <= 2^-24 relative, so the
 * rounding error is that of an fp32 convolution, at 2.67x the exact-fp32 MFMA rate.  Same
 * arguments and epilogue as ds_conv2d with ks = 3; weights are packed (and pre-split) by
 * ds_conv2d_x6_pack_weights into ds_conv2d_x6_packed_bytes(Cout, Cin) bytes. */
size_t ds_conv2d_x6_packed_bytes(int Cout, int Cin);
int ds_conv2d_x6_pack_weights(void* packed, const float* w, int Cout, int Cin, void* stream);
int ds_conv2d_x6(float* out, const float* in, const void* w_packed, const float* bias,
                 const float* shift, int shift_stride, const float* res1, const float* res2,
                 int B, int Cin, int Cout, int H, int W, int load_mode, void* stream);

/* The same convolution on the fp16 matrix cores ("fp16x3"): operands split into fp16 hi + lo
 * pieces (22-23 significand bits kept), three piece products accumulated in fp32; weights are
 * pre-scaled by 2^wshift at pack time (undone exactly in the epilogue) so that their low pieces
 * stay normal; gfx950's fp16 MFMA honours subnormal inputs.  Representation error ~1e-7 relative,
 * i.e. below the accumulation-order noise of an fp32 convolution.  Twice the rate of ds_conv2d_x6.
 * Domain.  fp16 has 5 exponent bits: x = hi + lo keeps its 22 bits for |x| in [2^-3, 2^16), degrades below (absolute
 * floor 2^-25) and overflows above.  Inputs that are normalised by construction (the prenorm loader, the *_img entry
 * points) sit inside that window.  Any other input -- the reference's fp32 convolutions take raw user fields
 * (punetg.py:719-735) and c_in = 1 parameterisations (preconditioners.py:139-161) at any magnitude -- is given with
 *   in_amax  [B] per-sample max |in| as float bits (NULL: no scaling): the loader multiplies sample b by 2^k, k chosen
 *            so that its maximum lands in [2^13, 2^14), and the epilogue undoes it exactly (2^-(wshift + k)): a block
 *            floating point with the sample's exponent, the whole fp32 range of magnitudes, results of a sample
 *            independent of the rest of the batch.  Produced by
 *   out_amax [B] (NULL: off): the epilogue merges max |out[b]| into slot b with atomicMax -- zero the slots first
 *            (ds_fill_u32) -- or by ds_absmax_rows for tensors that come from elsewhere. */
size_t ds_conv2d_h3_packed_bytes(int Cout, int Cin);
int ds_conv2d_h3_pack_weights(void* packed, const float* w, int Cout, int Cin, int wshift, void* stream);
int ds_conv2d_h3(float* out, const float* in, const void* w_packed, int wshift, const float* bias,
                 const float* shift, int shift_stride, const float* res1, const float* res2,
                 int B, int Cin, int Cout, int H, int W, int load_mode,
                 const float* prenorm, float* tile_stats, const unsigned* in_amax, unsigned* out_amax, void* stream);
/* out[r] = max(out[r], bits(max |x[r*row_stride .. +n_per_row)|)): the in_amax of a tensor no epilogue produced (one read pass,
 * HBM-bound).  Merging semantics: zero the slots first; several calls accumulate (channel concatenations). */
int ds_absmax_rows(unsigned* out, const float* x, int rows, size_t n_per_row, size_t row_stride, void* stream);
int ds_amax_merge(unsigned* out, const unsigned* a, const unsigned* b /* or NULL */, int n, void* stream);   /* out[i] = max(out[i], a[i], b[i]):
                                                                       the amax of a channel concatenation (adm.py:299) from those of its parts */
int ds_fill_u32(unsigned* p, unsigned value, size_t n, void* stream);          /* zeroing amax slots: a kernel, not a memset node (ds_amax.hip) */
/* The same reduction for an INPUT layer, x [B, C, HW] (user data: c_in * x next to raw condition fields, punetg.py:719-735):
 * scratch [B*C] (zeroed) receives the per-(sample, channel) maxima, out[b] their maximum m, and *flag (may be NULL) is OR-ed
 * with 1 when one exponent per sample cannot serve the layer: with wmax [C] (the layer's largest |weight| per input channel)
 * when m * wmax[c] > 2^gap * max_c'(max|x[b,c']| * wmax[c']) for a channel c that carries data -- the common exponent's
 * rounding error, seen through that channel's weights, would exceed fp32's own accumulation noise (a field of 1e-8 whose
 * weights are 1e8 times the others'); without wmax when a non-zero channel lies more than `gap` binades below m.  The host then
 * re-runs the input layer on the exact-fp32 kernel (nets/precision.py). */
int ds_absmax_channels(unsigned* out, unsigned* flag, unsigned* scratch, const float* x, const float* wmax, int B, int C,
                       size_t HW, int gap, void* stream);
/* The same in ONE launch for inputs of moderate size (C <= 64; one workgroup per sample): also zeroes the forward's amax arena
 * [arena_rows][B] (every row but out_row) -- a network evaluation starts with this call instead of ds_fill_u32 + ds_absmax_channels. */
int ds_input_amax(unsigned* arena, int arena_rows, int out_row, unsigned* flag, const float* x, const float* wmax, int B, int C,
                  size_t HW, int gap, void* stream);
/* Two optional fusions of the normalisation around the convolution (NULL = off):
 *   prenorm    [B, ceil16(Cin), 4] = (M, A, C, -), rows past Cin zero: the loader applies SiLU((x - M)*A + C) to every input element
 *              before the convolution (zero padding stays zero) -- the norm -> act of ResnetBlockC /
 *              ADMBaseBlock (commonlayers.py:824-829, adm.py:312-337) without materialising its output.
 *              Not with MAXPOOL2.  Tables come from ds_inorm_table / ds_gnorm1_table; their fourth column carries the sample's
 *              activation exponent (the loader produces SiLU(.) * 2^k), so in_amax is not used with prenorm.
 *   tile_stats [B, Cout, ntiles, 4]: per output channel and pixel tile of the stored values, (K, S, Q, n):
 *              n valid pixels, K one of them, S = sum(x-K), Q = sum((x-K)^2) (shifted sums: no mean^2
 *              cancellation in fp32); ntiles = ds_conv_tile_count(H, W); consumed by the *_table calls,
 *              which recombine the tiles in fp64.  16-byte aligned. */
int ds_conv_tile_count(int H, int W);

/* "Nearest x2 upsampling, then the 3x3 convolution" (UpSampler, commonlayers.py:145; ADM up blocks, adm.py:312-323)
 * evaluated at the LOW resolution: for each of the four output parities the 3x3 kernel collapses to a 2x2 kernel
 * on the un-upsampled input (taps summed at pack time), 16 instead of 36 multiply-adds per output.  in is
 * [B,Cin,Hl,Wl], out [B,Cout,2Hl,2Wl]; same epilogue terms, prenorm and tile_stats (4 tiles per low-resolution
 * tile: the count equals ds_conv_tile_count(2Hl, 2Wl)) as ds_conv2d_h3.  flags: DS_PAD_CIRCULAR and/or
 * DS_RES1_UPSAMPLED.  Only for inputs that are whole 8x32 or 16x16 tiles (ds_conv2d_h3_up_supported); other
 * shapes use ds_conv2d_h3 with DS_LOAD_UPSAMPLE2. */
/* The same convolution with its input given as pre-split fp16 hi / lo IMAGES in the kernel's LDS layout -- what
 * ds_inorm_silu_images writes: [B][ceil(Cin/16)][piece 2][half 2][H+2][W+2] 16-byte vectors of 8 channels, zero border (the
 * convolution's zero padding), zero channels past Cin.  Patches are staged by LDS-DMA: no staging registers, no split in the
 * kernel (16-29 % of ds_conv2d_h3; with several channel tiles every element used to be split once per tile).  Plain load, zero
 * padding, no fused norm; Cin must give an even number of 16-channel chunks (DS_ERR_UNSUPPORTED otherwise).  Epilogue terms and
 * tile_stats as ds_conv2d_h3.  Replaces conv(SiLU(norm(x))) of commonlayers.py:824-833 where the norm is not folded. */
/* The producer of those images: ds_inorm_silu (GroupNorm(C,C) / GroupRMSNorm(C,C) + SiLU, commonlayers.py:766-770, 372-384, 824,
 * 829) with the result written pre-split instead of as fp32 -- the same 8 bytes per element, values bit-identical to ds_inorm_silu's for planes of up to 1024 floats (beyond that the statistics are summed in
 * another order).  Planes of H*W <= 4096 floats, H*W a multiple of 4: ds_inorm_silu_images_supported. */
int ds_inorm_silu_images_supported(int H, int W);
int ds_inorm_silu_images(void* images, const float* x, const float* w, const float* b, int B, int C, int H, int W, float eps,
                         int kind, void* stream);
size_t ds_conv_images_bytes(int B, int C, int H, int W);
int ds_conv2d_h3_img(float* out, const void* images, const void* w_packed, int wshift, const float* bias, const float* shift,
                     int shift_stride, const float* res1, const float* res2, int B, int Cin, int Cout, int H, int W,
                     int flags /* 0 or DS_RES1_UPSAMPLED */, float* tile_stats, unsigned* out_amax, void* stream);
/* ds_gnorm1_apply (ADM's GroupNorm(1,C) / GroupRMSNorm(1,C) [+FiLM] + SiLU [+AvgPool2d(2)], adm.py:306-343) writing those images. */
int ds_gnorm1_apply_images(void* images, const float* x, const float* stats, const float* w, const float* b,
                           const float* film_scale, const float* film_shift, int film_stride, int B, int C, int Ho, int Wo,
                           int kind, int pool, void* stream);

/* The fused loader's SiLU((x - M)*A + C) from a norm table [B, ceil16(C), 4] (ds_inorm_table / ds_gnorm1_table), written as
 * those images: any plane size, every norm the tables describe (commonlayers.py:824-829 where ds_inorm_silu_images does not apply). */
int ds_table_apply_images(void* images, const float* x, const float* table, int B, int C, int H, int W, void* stream);

int ds_conv2d_h3_up_supported(int Hl, int Wl);
size_t ds_conv2d_h3_up_packed_bytes(int Cout, int Cin);
int ds_conv2d_h3_up_pack_weights(void* packed, const float* w, int Cout, int Cin, int wshift, void* stream);
int ds_conv2d_h3_up(float* out, const float* in, const void* w_packed, int wshift, const float* bias,
                    const float* shift, int shift_stride, const float* res1, const float* res2,
                    int B, int Cin, int Cout, int Hl, int Wl, int flags, const float* prenorm, float* tile_stats,
                    const unsigned* in_amax, unsigned* out_amax, void* stream);
/* ds_conv2d_h3_up (zero padding, no fused norm) with the LOW-resolution input given as pre-split images
 * (ds_gnorm1_apply_images / ds_inorm_silu_images over [B, Cin, Hl, Wl]); w_packed from ds_conv2d_h3_up_pack_weights.
 * ADM's norm1 -> SiLU -> nearest x2 -> conv1 of an 'up' block (adm.py:312-323). */
int ds_conv2d_h3_up_img(float* out, const void* images, const void* w_packed, int wshift, const float* bias,
                        const float* shift, int shift_stride, const float* res1, const float* res2,
                        int B, int Cin, int Cout, int Hl, int Wl, float* tile_stats, unsigned* out_amax, void* stream);


/* PUNetG norms from tile statistics: table [B, ceil16(C), 4]; table[b,c] = (mean | 0, rstd*w[c], b[c], 2^-k) for GroupNorm(C,C)
 * (kind 0) / GroupRMSNorm(C,C) (kind 1), (0, 1, 0, 2^-k) for no normalisation (kind 2); count = H*W.
 * commonlayers.py:766-770, 372-384, 891-899. */
int ds_inorm_table(float* table, const float* tile_stats, const float* w, const float* b, int B, int C, int ntiles,
                   int count, float eps, int kind, void* stream);
/* The fourth column (both table calls): 2^-k, the sample's activation exponent, the same in every row of the sample (padding
 * rows included); 0 in a hand-made table means none.  U = max_c |A_c| sqrt(n m2_c) + |C_c| bounds every |(x - M)*A + C| of the sample
 * (|x - mean| <= sqrt(n var), |x| <= sqrt(sum x^2)), k puts U at 2^13, the consuming loader produces SiLU(.) * 2^k at no cost (the
 * factor rides in the SiLU's own fma and reciprocal) and the convolution's epilogue undoes it: SiLU(norm(x)) stays inside the
 * fp16x3 window whatever the affine parameters, FiLM rows or eps-dominated variances do (commonlayers.py:766-770:
 * (x - mean)/sqrt(var + 1e-5) of a tensor of rms 1e-7 is 3e-5, not 1). */

/* ADM norms from tile statistics, per sample over (C, H, W), optionally over the channel concatenation
 * of two tensors (Cb = 0: one source): kind 0 GroupNorm(1,C): (mean_b, rstd_b*w[c], b[c]); kind 1
 * GroupRMSNorm(1,C): (0, w[c]/d_b, b[c]); with FiLM rows (both or neither NULL; the block's second norm, whichever
 * kind): A *= scale[b,c], C = C*scale[b,c] + shift[b,c].  count = C*H*W.
 * adm.py:306-343, 385-406, 764-766. */
int ds_gnorm1_table(float* table, const float* stats_a, int Ca, int ntiles_a, const float* stats_b, int Cb,
                    int ntiles_b, const float* w, const float* b, const float* film_scale, const float* film_shift,
                    int film_stride, int B, long long count, float eps, int kind, void* stream);
/* ds_gnorm1_stats without a pass over the tensor: the pairs ds_gnorm1_apply / ds_gnorm1_apply_images take, (mean, rstd)
 * (kind 0) or (0, RMS denominator) (kind 1), recombined in fp64 from the producers' tile statistics as ds_gnorm1_table does. */
int ds_gnorm1_stats_tiles(float* stats, const float* stats_a, int Ca, int ntiles_a, const float* stats_b, int Cb,
                          int ntiles_b, int B, long long count, float eps, int kind, void* stream);

/* 3x3 "same" convolution for Cout <= 4 (the networks' output layers: punetg.py:415, adm.py:193-215) as an
 * exact-fp32 FMA chain, streaming the input once (HBM-bound) instead of padding Cout to a 64-channel MFMA
 * tile.  w: torch layout [Cout, Cin, 3, 3] (no repacking).  circular != 0: periodic padding. */
int ds_conv2d_direct(float* out, const float* in, const float* w, const float* bias, int B, int Cin, int Cout,
                     int H, int W, int circular, void* stream);

/* 3x3x3 'same' convolution of volumes [B,Cin,D,H,W] -> [B,Cout,D,H,W] (PUNetG dimension = 3: Conv3d, CircularConv3d,
 * MagnitudePreservingConv3d -- commonlayers.py:25-160,973-1040; normedlayers.py:58-92), exact fp32, torch weight
 * layout [Cout,Cin,3,3,3].  load_mode: DS_LOAD_PLAIN, DS_LOAD_MAXPOOL2 (in is [B,Cin,2D,2H,2W]: DownSampler's
 * MaxPool3d(2) in the loader) or DS_LOAD_UPSAMPLE2 (in is [B,Cin,D/2,H/2,W/2]: UpSampler's nearest x2), optionally
 * OR-ed with DS_PAD_CIRCULAR.  out = (((conv + bias[co]) + shift[b,co]) + res1) + res2. */
int ds_conv3d_direct(float* out, const float* in, const float* w, const float* bias, const float* shift,
                     int shift_stride, const float* res1, const float* res2, int B, int Cin, int Cout, int D, int H,
                     int W, int load_mode, void* stream);

/* Volumes on the matrix cores: out[b,:,z] = sum_kz conv2d(in[b,:,z+kz-k/2], w[:,:,kz]) -- k launches of the 2-D
 * fp16x3 kernels per k x k x k convolution (three for 3x3x3) -- on a slice-major, depth-padded copy S[b][zp][c][y][x]
 * (zp = z + pad of D + 2 pad slices, pad = k/2; pads zero, or the wrapped neighbours when circular), in which a depth slice is
 * a 2-D sample and a depth tap a pointer offset of one slice (C*HW floats).
 *   ds_volume_to_slices: x [B,C,Din,HW] -> S [B,D+2,C,HW]; depth_mode 0 copy (Din = D), 1 max of depth pairs (Din = 2D:
 *     the depth half of MaxPool3d(2); the 2-D loader's MAXPOOL2 does H, W), 2 nearest x2 in depth (Din = D/2).
 *   ds_slices_to_volume: y [B,C,D,HW] = S interior (+ res1 + res2, volume layout).
 * The caller runs ds_conv2d_h3 on batch B*(D+2) - 2, in = S_in + (1 + dz)*Cin*HW_in, out = S_out + Cout*HW,
 * accumulating the second and third tap through res1 = out. */
int ds_volume_to_slices(float* slices, const float* x, int B, int C, int D, size_t HW, int depth_mode, int circular,
                        int pad /* 0..3 */, void* stream);
int ds_slices_to_volume(float* y, const float* slices, const float* res1, const float* res2, int B, int C, int D,
                        size_t HW, int pad, void* stream);

/* The same copies with the residual block's normalisation folded in (commonlayers.py:824-833 on volumes):
 *   ds_volume_to_slices_act: S = SiLU((x - M) A + C) with (M, A, C) = table[b][c] (ds_inorm_table rows, [B, ceil16(C), 4]);
 *     pad slices zero, or (circular) the activated wrapped neighbours -- the block's first GroupNorm / GroupRMSNorm + SiLU
 *     without a pass of its own.
 *   ds_slices_to_volume_stats: ds_slices_to_volume that also leaves the shifted partial sums (K, S, Q, n) of the stored values,
 *     stats [B, C, ds_volume_stat_tiles(D, HW), 4], for ds_inorm_table (count = D*HW) -- the NEXT block's first norm.
 *   ds_slice_tables: for a volume that stays slice-major between the two convolutions of a block: the tile statistics of the
 *     last depth-tap launch (ds_conv2d_h3's tile_stats over the 2-D batch B*(D+2) - 2, sample j = slice j + 1) -> one
 *     (M, A, C) row per (slice, channel), table [B*(D+2), ceil16(C), 4]; pad slices get zero rows, so the consumer's fused
 *     loader (prenorm = table + (1 + dz) rows) reads SiLU(0) = 0 there: the depth axis' zero padding.  count = D*HW.
 *     circular: the pad slices get the sample's row too, and the caller fills them with wrapped copies first:
 *   ds_wrap_pad_slices: S [B, D+2, C, HW] produced slice-major (interior written by the tap launches): pad slice 0 of every
 *     sample <- its slice D, pad slice D+1 <- its slice 1 -- the periodic depth padding (commonlayers.py:918-971 on volumes). */
int ds_volume_stat_tiles(int D, size_t HW);
int ds_volume_to_slices_act(float* slices, const float* x, const float* table, int B, int C, int D, size_t HW, int circular,
                            void* stream);
int ds_wrap_pad_slices(float* slices, int B, int C, int D, size_t HW, void* stream);
int ds_slices_to_volume_stats(float* y, const float* slices, const float* res1, const float* res2, float* stats, int B, int C,
                              int D, size_t HW, int pad, void* stream);
int ds_slice_tables(float* table, const float* tile_stats, const float* w, const float* b, int B, int C, int D, int ntiles,
                    long long count, float eps, int kind, int circular, void* stream);

/* Resampling of volumes in ADM blocks with dimension = 3 (make_downsample / make_upsample, adm.py:352-384):
 *   ds_avgpool3d:  AvgPool3d(2): x [planes, 2Do, 2Ho, 2Wo] -> out [planes, Do, Ho, Wo] (sum of the 8 voxels in (z, y, x) order, / 8);
 *   ds_upsample3d: Upsample(scale_factor=2, mode='nearest'): x [planes, Di, Hi, Wi] -> out [planes, 2Di, 2Hi, 2Wi].
 * planes = B*C. */
int ds_avgpool3d(float* out, const float* x, int planes, int Do, int Ho, int Wo, void* stream);
int ds_upsample3d(float* out, const float* x, int planes, int Di, int Hi, int Wi, void* stream);

/* 1x1 convolution in the fp16x3 scheme of ds_conv2d_h3 (same epilogue terms, same domain and
 * in_amax / out_amax; amax_split > 0 (a multiple of 64): channels >= amax_split report to out_amax[B + b] instead of
 * out_amax[b] -- the attention in-projection keeps one exponent for q and k and one for v).  ADM's residual projection convresidual(resample(x)) (adm.py:345-349) with the
 * resampling folded into the load: load_mode PLAIN, UPSAMPLE2 (nearest x2, in is [B,Cin,H/2,W/2])
 * or AVGPOOL2 (AvgPool2d(2) in torch's summation order, in is [B,Cin,2H,2W]); also the attention
 * in/out projections.  Packed weights: [ceil(Cout/64)][ceil(Cin/16)][piece 2][h 2][64 co][8 ci] fp16. */
size_t ds_conv1x1_h3_packed_bytes(int Cout, int Cin);
int ds_conv1x1_h3_pack_weights(void* packed, const float* w, int Cout, int Cin, int wshift, void* stream);
int ds_conv1x1_h3(float* out, const float* in, const void* w_packed, int wshift, const float* bias,
                  const float* shift, int shift_stride, const float* res1, const float* res2,
                  int B, int Cin, int Cout, int H, int W, int load_mode, float* tile_stats,
                  const unsigned* in_amax, unsigned* out_amax, int amax_split, void* stream);

/* Single-head self-attention over L = H*W positions, channel-major operands:
 *   qkv [B, 3E, L] (rows 0..E-1 = Q^T, E..2E-1 = K^T, 2E..3E-1 = V^T), out [B, E, L] = (softmax(Q K^T / sqrt(E)) V)^T.
 * nn.MultiheadAttention(E, num_heads=1) core, attention.py:41-43,67.  L a multiple of 32; E in {32, 64, 128, 256, 384, 512}. */
int ds_attention(float* out, const float* qkv, int B, int E, int L, void* stream);

/* The same contract for any E and L (one wave per query, exact fp32 FMA chains): the path for sequence
 * lengths that are not a multiple of 32 (tiny or odd bottlenecks, e.g. the reference's own 16x16 ADM test). */
int ds_attention_generic(float* out, const float* qkv, int B, int E, int L, void* stream);

/* Cosine attention (attn_type="cosine": cosine_product_attn / cosine_similarity, attention.py:300-372): queries and
 * keys are divided by (their per-token L2 norm over the channels + eps) before the product, and the logits carry no
 * 1/sqrt(E).  In place on x [B, Ctot, L], channels [c0, c0+C): x = x/(||x_token|| + eps)*gain.  Call it on the q
 * block of qkv with gain = sqrt(E) (cancelling the attention kernels' 1/sqrt(E)) and on the k block with gain = 1. */
int ds_token_l2_normalize(float* x, int B, int Ctot, int c0, int C, int L, float eps, float gain, void* stream);

/* The same attention with both matrix products on the fp16 matrix cores in the fp16x3 scheme of
 * ds_conv2d_h3 (operands split into fp16 hi + lo, three products, fp32 accumulation and fp32
 * softmax statistics): fp32-level accuracy for |q|, |k|, |v| < 65504 and not far below 1, or -- with in_amax [2][B], the
 * per-sample max |q, k| and max |v| the in-projection left (ds_conv1x1_h3 with amax_split = 2E) -- for any magnitude: q and k
 * are staged times 2^kqk, v times 2^kv, the logits read back times 2^-2kqk, the output times 2^-kv.  out_amax [B]: max |out[b]|,
 * merged.  Same layouts; E <= 256. */
int ds_attention_h3(float* out, const float* qkv, int B, int E, int L, const unsigned* in_amax, unsigned* out_amax, void* stream);
/* The same with a caller-provided workspace of ds_attention_h3_workspace_bytes(B, E, L) bytes (16-byte aligned): a
 * pre-pass splits K and V once per sample into images that have the kernel's LDS layout, and the attention kernel
 * stages its key tiles by LDS-DMA instead of re-splitting them in every 128-query workgroup -- the form for long
 * sequences (L = 4096 tokens at a 64 x 64 bottleneck: every tile is otherwise split 32 times).  Same results bit for bit. */
size_t ds_attention_h3_workspace_bytes(int B, int E, int L);
int ds_attention_h3_ws(float* out, const float* qkv, void* workspace, int B, int E, int L, const unsigned* in_amax,
                       unsigned* out_amax, void* stream);

/* y[m, n] = act(sum_k x[m,k]*w[n,k] + b[n]); act 0 none, 1 SiLU, 2 ReLU.  torch Linear layout.
 * ResnetTimeBlock (commonlayers.py:516-522) and MLPUncond (mlp.py:30-37). b may be NULL. */
int ds_linear(float* y, const float* x, const float* w, const float* b, int M, int K, int N, int act, void* stream);

/* out[m, j] = sin(2*pi*t[m]*W[j]), out[m, half+j] = cos(...).  GaussianFourierProjection.forward,
 * commonlayers.py:185-190.  add (optional, [m_add, 2*half], m_add in {1, M}) is added afterwards
 * (te + cond_dropout(ye), punetg.py:410). */
int ds_fourier_features(float* out, const float* t, const float* W, const float* add, int add_rows,
                        int M, int half, void* stream);

/* ConvolutionalFourierProjection.forward (commonlayers.py:246-255; PUNetG's convin when in_embedding=True, punetg.py:194-202):
 * out[b, d, p] = sin(sum_c x[b,c,p] * (2*pi*W[c,d])), out[b, D+d, p] = cos(same); x [B,C,HW], W [C,D], out [B,2D,HW].
 * (The reference layer only runs with bias=False -- its bias branch adds an int to a list -- in which case PUNetG
 * appends a constant-one input channel whose W row acts as the phase.) */
int ds_fourier_channels(float* out, const float* x, const float* W, int B, int C, int D, size_t HW, void* stream);

/* ------------------------------------------------------------------------------------
 * ADM score-network layers (adm.py) not shared with PUNetG.
 * ---------------------------------------------------------------------------------- */

/* Per-sample statistics over the whole (C, H, W) volume: GroupNorm(num_groups=1) / GroupRMSNorm(1, C),
 * adm.py:385-406.  stats[2b] = mean, stats[2b+1] = 1/sqrt(var_biased+eps) (kind 0), or
 * stats[2b] = 0, stats[2b+1] = sqrt(mean(x^2)+eps) (kind 1).  x: [B, C, HW].  workspace: device
 * scratch of ds_gnorm1_workspace_bytes(B) bytes (fp64 partial sums of the two-phase reduction). */
size_t ds_gnorm1_workspace_bytes(int B);
int ds_gnorm1_stats(float* stats, void* workspace, const float* x, int B, int C, int HW, float eps, int kind,
                    void* stream);

/* One elementwise pass fusing the normalisation with what follows it in ADMBaseBlock (adm.py:306-343):
 *   kind 0: SiLU(n) or SiLU(n*scale[b,c] + shift[b,c]), n = (x-mean)*rstd*w[c]+b[c]     GroupNorm(1, C)
 *   kind 1: likewise with n = x/denom*w[c]+b[c]                                          GroupRMSNorm(1, C)
 *           (FiLM when scale/shift are given: norm1 -> act, adm.py:327-328; norm2 -> FiLM -> act, adm.py:331,307,335;
 *           ADM's defaults are kind 0 without and kind 1 with FiLM, make_norm_layers adm.py:385-406 allows either)
 *   kind 2: x                                                    residual-branch input (adm.py:345-347)
 * then, pool = 1, the block's AvgPool2d(2) (adm.py:316-319).  x [B,C,H,W] -> out [B,C,H(/2),W(/2)].
 * scale/shift: rows of embed_linear(te), film_stride floats between samples (0 = shared row). */
int ds_gnorm1_apply(float* out, const float* x, const float* stats, const float* w, const float* b,
                    const float* film_scale, const float* film_shift, int film_stride,
                    int B, int C, int H, int W, int kind, int pool, void* stream);

/* out[b] = cat(a[b], b[b]) along channels (na, nb floats per sample): the decoder's skip concat,
 * adm.py:764-766. */
int ds_concat2(float* out, const float* a, const float* b, int B, size_t na, size_t nb, void* stream);

/* out[m,n] = act(a[m,n] + add[m or 0, n]); act as in ds_linear.  ADMTimeEmbedding's "te + ye" and
 * final SiLU (adm.py:1050-1052). */
int ds_add_act(float* out, const float* a, const float* add, int add_rows, int M, int N, int act, void* stream);

/* out[b, i] = x[b, i]*(1 - mask[i]) + y[b, i]*mask[i]: the known-region re-imposition of
 * Scheduler.inpaint / repaint (schedulers.py:112,116,146); mask has n_per_sample entries. out may alias x. */
int ds_mask_blend(float* out, const float* x, const float* y, const float* mask, size_t n_per_sample, int B,
                  void* stream);

/* Non-constant-scaling (VP) branch of Scheduler.rhs, schedulers.py:275-293:
 *   ds_div_scalar: out = x / s                 (the score is evaluated at x/s)
 *   ds_axpby:      out = a*x + b*y (y may be NULL: out = a*x);  (s'/s)*x - multiplier*score is
 *                  a = s'/s, b = -multiplier (the negation is exact). out may alias x or y. */
int ds_axpby(float* out, const float* x, float a, const float* y, float b, size_t n, void* stream);
int ds_div_scalar(float* out, const float* x, float s, size_t n, void* stream);

/* DimensionAgnosticBatchNorm with its running statistics (eval mode), around the sampling loop when
 * KarrasModuleConfig.has_edm_batch_norm (karrasmodule.py:1209-1210,1225-1226; aux_scripts/batchnorm.py:111-170):
 *   inverse = 0 (normalize):   ((x - mean)/sqrt(var + eps) [*weight + bias]) * sigma
 *   inverse = 1 (unnormalize): ((x/sigma [- bias)/weight]) * sqrt(var + eps) + mean
 * x, out [B, C, HW]; mean/var (and weight/bias, or both NULL) have nc = 1 or C entries. out may alias x. */
int ds_batchnorm_eval(float* out, const float* x, const float* mean, const float* var, const float* weight,
                      const float* bias, float eps, float sigma, int inverse, int B, int C, int nc, size_t HW,
                      void* stream);

/* out[i] = x1 + ((x2 - x1)*i)/(n - 1), i = 0..n-1, each of numel floats: linear_interpolation
 * (torchutils.py:64-65) used by KarrasModule.interpolate_images (karrasmodule.py:1136-1138). */
int ds_lerp_stack(float* out, const float* x1, const float* x2, int n, size_t numel, void* stream);

/* out = a + b (n floats). */
int ds_add(float* out, const float* a, const float* b, size_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * hipGraph capture of a launch sequence (the whole N-step loop is captured once per
 * (network, nsteps, batch) and replayed).
 * ---------------------------------------------------------------------------------- */
typedef struct ds_graph ds_graph;
int ds_graph_begin_capture(void* stream);
int ds_graph_end_capture(void* stream, ds_graph** out_graph, int* node_count);
int ds_graph_launch(ds_graph* g, void* stream);
int ds_graph_destroy(ds_graph* g);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSCI_HIP_H */
