/* sdmatte.h - C ABI of the MI355X-native SDMatte engine (libsdmatte_hip.so, gfx950 only).
 *
 * The reference (flybirdxx/ComfyUI-SDMatte) is pure Python and has no FFI; this header is the boundary a
 * maintainer binds with ctypes.  Each entry point names the reference interface it
 * replaces.  Plain pointers and sizes only - no torch types.  All functions return 0 on success or a
 * negative sdm_status; sdm_last_error() gives the message (the Python shim raises RuntimeError with it,
 * mirroring how exceptions propagate to ComfyUI in the reference).  Nothing here ever abort()s.
 *
 * Pointer kinds: every data pointer is either a HOST pointer or a DEVICE pointer of the engine's GPU,
 * selected by the `ptr_kind` argument (SDM_PTR_HOST / SDM_PTR_DEVICE).  Device pointers are what
 * `tensor.data_ptr()` returns for PyTorch-ROCm tensors at the node boundary.
 */
#ifndef SDMATTE_H_
#define SDMATTE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdm_ctx sdm_ctx;

enum sdm_status {
  SDM_OK = 0,
  SDM_ERR_INVALID = -1,   /* bad argument / shape */
  SDM_ERR_HIP = -2,       /* HIP runtime error */
  SDM_ERR_STATE = -3,     /* weights not loaded / finalised */
  SDM_ERR_NOMEM = -4,
  SDM_ERR_NODEVICE = -5   /* no gfx950 GPU visible: the product never falls back to the CPU */
};

enum sdm_ptr_kind { SDM_PTR_HOST = 0, SDM_PTR_DEVICE = 1 };
enum sdm_dtype { SDM_F32 = 0, SDM_F16 = 1, SDM_BF16 = 2 };

/* Architecture constants.  The reference reads them from stable-diffusion-2-1-base/{unet,vae}/config.json
 * (sdmatte_nodes.py:20-31, meta_arch.py:95-118) plus in-code defaults (meta_arch.py:107-112). */
typedef struct sdm_config {
  int32_t vae_channels[4];
  int32_t vae_layers_per_block;
  int32_t unet_channels[4];
  int32_t unet_heads[4];
  int32_t unet_layers_per_block;
  int32_t cross_attention_dim;
  int32_t unet_in_channels;
  int32_t unet_out_channels;
  int32_t bbox_embeddings_input_dim;
  int32_t groups;
  float vae_eps;
  float unet_res_eps;
  float unet_tf_gn_eps;
  float unet_ln_eps;
  float vae_scaling_factor;
  float attn_mask_value;
  int32_t stream_f32;      /* 1: residual stream tensors kept in fp32 (default), 0: fp16 */
  int32_t point_embeddings_input_dim;   /* 1680 (meta_arch.py:107-108); 0 = default */
  /* Arithmetic precision of the MFMA contractions, one bit per stage (enum sdm_precise_stage).  0 = fp16 operands everywhere
   * (fast: alpha within ~4e-3 of the fp32 reference path, the rounding floor of ANY fp16-operand evaluation).  A set bit evaluates
   * that stage with split operands (x = hi + lo with hi = fp16(x): 22 significant bits; hi.hi + lo.w + x.lo_w in fp32 accumulators)
   * and keeps every activation in fp32 between kernels: SDM_PRECISE_ALL reproduces the reference's fp32 CPU path to ~1.2e-4 (1e-3
   * is the parity bar, sdmatte_nodes.py:355-360) at roughly 2x the matrix-pipe time.  The residual terms (2^-11 of a product) run
   * on fp8 e4m3 operands in the wide 3x3 convs and the K >= 1024 GEMMs, on fp16 elsewhere; in the attention cores Q.K^T is split,
   * P.V is plain fp16. */
  int32_t precise_mask;
  int32_t reserved[5];
} sdm_config;

enum sdm_precise_stage {
  SDM_PRECISE_VAE_ENC = 1,        /* VAE encoder convs + quant_conv */
  SDM_PRECISE_VAE_DEC = 2,        /* post_quant_conv + VAE decoder convs */
  SDM_PRECISE_VAE_ATTN_LIN = 4,   /* q|k|v / to_out linears of the two VAE mid-block attentions */
  SDM_PRECISE_UNET_RES = 8,       /* U-Net conv_in/out, ResBlock convs, shortcuts, down/up-samplers */
  SDM_PRECISE_UNET_TF = 16,       /* U-Net transformer linears (proj_in/out, q|k|v, to_out, GEGLU, folded cross K|V) */
  SDM_PRECISE_UNET_ATTN = 32,     /* U-Net attention cores (QK^T, softmax, PV; head dim 64) */
  SDM_PRECISE_ALL = 63            /* (the single-head d=512 VAE attention core always runs on fp16 operands) */
};

/* Fill `cfg` with the SD-2.1-base / SDMatte constants and precise_mask = SDM_PRECISE_ALL (the precision that
 * meets the parity bar; set it to 0 for the fast fp16-operand graph). */
void sdm_default_config(sdm_config* cfg);

/* Create an engine on GPU `device_id`.  Replaces SDMatte.__init__ / init_submodule (meta_arch.py:31-124)
 * + `.to(device)` (sdmatte_nodes.py:323).  cfg == NULL selects sdm_default_config. */
int sdm_create(sdm_ctx** out, int device_id, const sdm_config* cfg);
void sdm_destroy(sdm_ctx* ctx);
const char* sdm_last_error(sdm_ctx* ctx);   /* ctx may be NULL: returns the last create() error */

/* Weight loading.  Replaces the safetensors read loop + load_state_dict(strict=False)
 * (sdmatte_nodes.py:300-321): call once per checkpoint tensor with its key (e.g.
 * "unet.down_blocks.0.resnets.0.conv1.weight"), dtype, shape and a HOST pointer (e.g. into the
 * safetensors mmap).  Unknown keys (text_encoder.*, point_embedding.*) are ignored and reported through
 * sdm_weight_stats; a shape mismatch is an error, as in torch.  Returns 1 if the key was consumed. */
int sdm_load_tensor(sdm_ctx* ctx, const char* name, int dtype, int ndim, const int64_t* shape, const void* host_ptr);
/* After the last tensor: folds constants and marks the engine ready.  n_missing = expected keys never
 * supplied (they stay zero-initialised; the reference would keep its random init). */
int sdm_finalize_weights(sdm_ctx* ctx);
int sdm_weight_stats(sdm_ctx* ctx, int64_t* n_loaded, int64_t* n_missing, int64_t* n_ignored);
/* Name of the i-th expected-but-missing key, or NULL. */
const char* sdm_missing_key(sdm_ctx* ctx, int64_t i);

/* Packed fp16 weight blob (for RCCL broadcast between ranks): size, and copy out/in to/from a
 * DEVICE buffer of that size.  Import marks every key as loaded; sdm_finalize_weights must follow. */
int64_t sdm_weight_blob_bytes(sdm_ctx* ctx);
int sdm_export_weight_blob(sdm_ctx* ctx, void* device_dst);
int sdm_import_weight_blob(sdm_ctx* ctx, const void* device_src);
/* Small host-side tensors needed for the time/bbox embedding constants (kept outside the blob). */
int64_t sdm_host_blob_bytes(sdm_ctx* ctx);
int sdm_export_host_blob(sdm_ctx* ctx, void* host_dst);
int sdm_import_host_blob(sdm_ctx* ctx, const void* host_src);

/* Model forward.  Replaces SDMatte.forward(data) (meta_arch.py:127-261):
 *   image  fp32 [B,3,S,S]  = data["image"]   (already resized + normalised to [-1,1])
 *   trimap fp32 [B,1,S,S]  = data["trimap"]  (in [-1,1])
 *   is_trans int32 [B]     = data["is_trans"];  coords fp32 [B,4] = data["trimap_coords"] (NULL -> [0,0,1,1])
 *   alpha  fp32 [B,1,S,S]  = return value in [0,1]
 * is_trans / coords are always HOST pointers (tiny).
 * Stream contract (all sdm_forward* / sdm_apply_matte): kernels run on a stream owned by the engine.  With SDM_PTR_DEVICE,
 * `stream` is the hipStream_t on which the caller produced the inputs and will consume the outputs - in PyTorch,
 * torch.cuda.current_stream().cuda_stream; NULL = the device's default stream.  The engine orders its work after everything
 * queued on that stream at call time (event wait, no host sync) and makes that stream wait for the outputs, so the call
 * behaves like any other kernel launch on `stream`.  With SDM_PTR_HOST the call copies in, runs, copies out and returns after
 * a host synchronisation; `stream` is ignored. */
int sdm_forward(sdm_ctx* ctx, const float* image_b3ss, const float* trimap_b1ss, int B, int S, const int32_t* is_trans,
                const float* coords_b4, float* alpha_b1ss, int ptr_kind, void* stream);

/* The other prompt types of the reference core (meta_arch.py:22-28,131-206): `aux` is data[aux_input]
 * ("trimap", "bbox_mask", "mask", "auto_mask" or "point_mask": [B,1,S,S] in [-1,1], encoded and used exactly like the trimap),
 * `cond` the matching coordinates: SDM_COND_BOX = data["*_coords"] [B,4] -> bbox_embedding (NULL -> [0,0,1,1], which is also
 * what use_coor_input=False feeds); SDM_COND_POINTS = data["point_coords"] [B,cond_dim] -> zero-padded to the first divisor of
 * point_embeddings_input_dim, sinusoid-embedded and sent through point_embedding (pass zeros for use_coor_input=False).
 * use_attention_mask = 0 runs the self-attention without the aux key mask (aux_input not in attn_mask_aux_input).
 * sdm_forward(...) == sdm_forward_ex(..., coords, 4, SDM_COND_BOX, 1, ...). */
enum sdm_cond_kind { SDM_COND_BOX = 0, SDM_COND_POINTS = 1 };
int sdm_forward_ex(sdm_ctx* ctx, const float* image_b3ss, const float* aux_b1ss, int B, int S, const int32_t* is_trans,
                   const float* cond, int cond_dim, int cond_kind, int use_attention_mask, float* alpha_b1ss, int ptr_kind, void* stream);

/* Rectangular inference (beyond the reference, whose attention-mask code asserts square latents,
 * replace.py:57-60): same as sdm_forward_ex with image [B,3,SH,SW], aux [B,1,SH,SW], alpha [B,1,SH,SW]; SH and SW multiples
 * of 64.  The level-k key bias keeps the reference's stride-2^k pick, bias_k[i,j] = bias_0[2^k i, 2^k j]. */
int sdm_forward_rect(sdm_ctx* ctx, const float* image_b3hw, const float* aux_b1hw, int B, int SH, int SW, const int32_t* is_trans,
                     const float* cond, int cond_dim, int cond_kind, int use_attention_mask, float* alpha_b1hw, int ptr_kind, void* stream);

/* Node-level call.  Replaces the device part of SDMatteApply.apply_matte (sdmatte_nodes.py:339-363):
 *   image fp32 [B,H,W,3] in [0,1], trimap fp32 [B,H,W] in [0,1]  ->  antialiased resize to SxS, normalise,
 *   forward, resize back to (H,W), clamp(0,1)  ->  alpha fp32 [B,H,W].
 * (alpha only; sdm_apply_matte_node below adds mask_refine and the output composition on the GPU.) */
int sdm_apply_matte(sdm_ctx* ctx, const float* image_bhwc, const float* trimap_bhw, int B, int H, int W, int S,
                    int is_transparent, float* alpha_bhw, int ptr_kind, void* stream);

/* The whole node in one call: sdm_apply_matte followed, at the original resolution and on the GPU, by
 * mask_refine (trimap_constraint; sdmatte_nodes.py:365-380) and the output composition (sdmatte_nodes.py:382-397):
 * output_mode 0 = alpha_only (matted = zeros [B,H,W,3]), 1 = matted_rgba ([B,H,W,4] = image | alpha), 2 = matted_rgb
 * ([B,H,W,3] = image gated by (trimap > 0.2) & (alpha > 0.1)).  Bit-identical to the reference's CPU tensor arithmetic. */
/* The trimap [B,trimap_h,trimap_w] is resized to SxS on its own, as in the reference (sdmatte_nodes.py:212-214,349): it only has to
 * match the image where the reference indexes the alpha with it, i.e. with mask_refine != 0 or output_mode 2 (SDM_ERR_INVALID otherwise).
 * trimap_constraint is a double: the thresholds are float32(c) and float32(1.0 - c) with 1.0 - c evaluated in double, as torch does
 * for the reference's Python-float comparisons. */
int sdm_apply_matte_node(sdm_ctx* ctx, const float* image_bhwc, const float* trimap_bhw, int B, int H, int W, int trimap_h, int trimap_w, int S,
                         int is_transparent, int output_mode, int mask_refine, double trimap_constraint, float* alpha_bhw, float* matted_bhwc,
                         int ptr_kind, void* stream);

/* Memory the engine holds outside any framework allocator: packed weights + activation arena (sized by the largest batch /
 * resolution seen) + I/O staging.  sdm_release_memory frees everything but the weights (the next forward re-allocates). */
int64_t sdm_resident_bytes(sdm_ctx* ctx);
/* The weight part of it: the canonical blob (sdm_weight_blob_bytes) plus the kernel-specific layouts derived from it.
 * sdm_resident_bytes - sdm_weight_bytes = what sdm_release_memory gives back. */
int64_t sdm_weight_bytes(sdm_ctx* ctx);
int sdm_release_memory(sdm_ctx* ctx);

/* Kernel-selection options.  The library reads NO environment variable: every choice among its kernel variants has one default, and
 * this is the only way to change one (tests and the A/B tools under tools/ do; the ComfyUI node never does).  Process-wide; names and
 * meanings: sdm_option_name(i) / sdm_option_help(i) for i = 0 .. until NULL.  Options marked "read when a model is built" / "read at
 * sdm_create" must be set before that call.  Returns SDM_ERR_INVALID for an unknown name. */
int sdm_set_option(const char* name, int value);
int sdm_get_option(const char* name, int* value);
void sdm_reset_options(void);
const char* sdm_option_name(int i);
const char* sdm_option_help(int i);
/* Which kernel variants were launched since the last reset, as "name=count;..." (returns the full length; truncates to cap).  Lets a
 * test assert that the variant it means to check is the one that ran. */
int sdm_kernel_counts(char* buf, int cap);
void sdm_kernel_counts_reset(void);

/* Block until everything queued on the engine stream has finished. */
int sdm_synchronize(sdm_ctx* ctx);

/* Time (ms) spent by the GPU in the last sdm_forward/sdm_apply_matte, measured with HIP events on the
 * stream the kernels were launched on.  Valid after sdm_synchronize. */
float sdm_last_forward_ms(sdm_ctx* ctx);

/* Per-kernel-class event timing of the next forward (debug/bench): enable, run, then read back
 * `n` (name, ms, launches) triples.  Adds an event pair per launch - not for the timed bench loop. */
int sdm_profile_enable(sdm_ctx* ctx, int on);
int sdm_profile_count(sdm_ctx* ctx);
/* CSV (kernel,ms,gflop,mbytes,desc), one line per launch of the last profiled forward. */
const char* sdm_profile_dump(sdm_ctx* ctx);
int sdm_profile_get(sdm_ctx* ctx, int i, const char** name, float* ms, int64_t* launches, double* flops, double* bytes);

/* ---- single-operator entry points (parity tests call the same kernels the engine uses) --------------
 * All pointers are DEVICE pointers.  Activations are NHWC; fp16 unless the *_f32 flag says otherwise. */

/* conv3x3 (ntaps=9) or 1x1/linear (ntaps=1): y = conv(concat(in0,in1)) [*scale] [+bias] [+res] | GEGLU.
 * w: fp32 OIHW [O][I][kh][kw] or [O][I]; I = (C0+C1) real channels.  stride 1|2; pad_mode 0: symmetric pad 1,
 * 1: VAE asymmetric (0,1,0,1) (stride 2).  up=1 fuses a nearest x2 upsample.  tile_cfg = -1 picks
 * automatically, >= 0 forces one of the compiled tile configurations (see sdm_conv_num_cfgs). */
int sdm_op_conv(sdm_ctx* ctx, const void* in0, const void* in1, int C0, int C1, int in_f32, int N, int Hin, int Win, int up,
                int stride, int pad_mode, int ntaps, const float* w, const float* bias, int O, void* out, int out_f32,
                const void* res, int res_f32, int geglu, float out_scale, int tile_cfg);
int sdm_conv_num_cfgs(int ntaps, int stride);
/* The same with (a) split != 0: split-fp16 operands (the precise mode's kernels; fp32 activations only) and (b) gn_gamma != NULL:
 * GroupNorm(gn_groups, eps)(+SiLU) of the input applied inside the conv's operand staging - the production path of every
 * ResnetBlock2D conv (3x3, stride 1; tile_cfg 0, 4 or 5). */
int sdm_op_conv_ex(sdm_ctx* ctx, const void* in0, const void* in1, int C0, int C1, int in_f32, int N, int Hin, int Win, int up,
                   int stride, int pad_mode, int ntaps, const float* w, const float* bias, int O, void* out, int out_f32,
                   const void* res, int res_f32, int geglu, float out_scale, int tile_cfg, int split, const float* gn_gamma,
                   const float* gn_beta, float gn_eps, int gn_groups, int gn_silu);
/* bench only (tools/gemm_p3_bench.py): ms per launch of the plane-fed GEMM on random operands; epi_flags = epilogue (0 fp32, 1 GEGLU, 2 q|k|v planes, 3 planes,
 * 4 fp32 + statistics) | 256 for an fp32 residual */
float sdm_bench_gemm_p3(sdm_ctx* ctx, long M, int K, int O, int epi_flags, int iters);
/* Plane-fed GEMM of the transformer blocks' Linear layers (k_gemm.h; reference call sites replace.py:232-362 -> diffusers BasicTransformerBlock) as a
 * stand-alone operator.  x: fp32 [N*H*W][K] on the device, K % 32 == 0; converted to the kernel's operand planes by the conversion kernel or, when
 * ln_gamma != NULL, by LayerNorm(eps) with plane output.  w: fp32 [O][K], packed exactly as a model layer.  mode 0: fp32 [rows][O] (+bias, +fp32 residual);
 * 1: GEGLU, O = 2 x outputs, result planes decoded to fp32 [rows][O/2]; 3: linear (+residual) to planes, decoded to fp32; 2: the raw q | k | v operand planes of
 * the attention cores (fp16 [rows][O], then the e5m2 pair plane of the same size; pair plane for channels < lo_cols only); 4: mode 0 + the per-(image, row
 * block, channel) {sum, sumsq} rows of the consumer's GroupNorm into `stats` ([N][*srows][O][2] floats; size it for 2 * ceil(H*W / 64) rows). */
int sdm_op_gemm_p3(sdm_ctx* ctx, const float* x, int N, int H, int W, int K, const float* w, const float* bias, int O, int mode, const float* res,
                   const float* ln_gamma, const float* ln_beta, float ln_eps, int lo_cols, void* out, float* stats, int* srows);
/* Test hooks for the exact algebraic folds done at load time (cross-attention K|V fold of aux_conv_in, logit scale in to_q,
 * time/opacity/bbox embedding constants in the conv1 bias tables): run one packed layer by name on an fp32 NHWC input
 * (DEVICE pointers; channel count = the layer's padded input channels), and read one folded bias row (HOST output). */
/* Test hook: class plane ([N][Hin][Win] bytes on the device; 0 = nothing known, 1..4 = region class) of the input of the NEXT sdm_op_conv_ex call.  The conv
 * then treats its input as the VAE encoder treats the trimap images ("constant tiles"): output tiles inside one region are filled, not multiplied. */
int sdm_debug_set_input_cmask(sdm_ctx* ctx, const unsigned char* mask);
int sdm_debug_run_layer(sdm_ctx* ctx, const char* layer_name, const float* x_nhwc, int N, int H, int W, float* out_nhwc, int Cout);
int sdm_debug_temb_row(sdm_ctx* ctx, int temb_index, int is_trans, const float* coords4, float* out_host, int cout);
/* Bench/ablation helper: ms per launch of one conv (random-ish data), HIP-event timed on the engine stream. */
float sdm_bench_attn(sdm_ctx* ctx, int B, int heads, int Lq, int Lk, int qt, int ablate, int iters);
float sdm_bench_conv(sdm_ctx* ctx, int N, int H, int W, int Cin, int Cout, int ntaps, int stride, int in_f32, int tile_cfg, int ablate, int iters);
/* GroupNorm(groups)+optional SiLU over NHWC (concat of two sources) -> fp16 NHWC. */
int sdm_op_groupnorm(sdm_ctx* ctx, const void* in0, const void* in1, int C0, int C1, int in_f32, int N, int HW, int groups,
                     const float* gamma, const float* beta, float eps, int silu, void* out_f16);
/* LayerNorm over the last dim C of [rows, C] -> fp16. */
int sdm_op_layernorm(sdm_ctx* ctx, const void* x, int in_f32, long rows, int C, const float* gamma, const float* beta,
                     float eps, void* out_f16);
/* softmax(q k^T * scale + bias) v per (batch, head): q [B,Lq,heads*D], k,v [B,Lk,heads*D] fp16 with row
 * strides ldq/ldk/ldv, bias fp32 [B,Lk] or NULL (natural-log domain, as in the reference), out [B,Lq,heads*D].
 * D = 64 (any heads) or 512 (heads = 1).  With a bias (D = 64), 64-key tiles in which every key's bias lies more than
 * 2000*ln(2) below the image's largest bias are not loaded: their probabilities underflow to exactly 0 in fp32, as they do in
 * the reference's softmax (trimap keys carry (1-m)*-10000, replace.py:401-403).  The result is bit-identical to walking
 * every tile; the engine option attn_dense = 1 disables the skip. */
int sdm_op_attention(sdm_ctx* ctx, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* bias,
                     int B, int heads, int Lq, int Lk, int D, void* out, int ldo);
/* Split-precision attention cores (head dim 64) as the default precision runs them: contiguous fp32 q [B,Lq,heads*64], k / v [B,Lk,heads*64]
 * are split into the operand planes the engine's GEMM epilogues produce (fp16 high parts + e5m2 residual pairs for Q.K^T, fp16 V), the
 * logit scale goes into Q; fp32 output [B,Lq,heads*64].  Test hook for the kernel the engine runs. */
int sdm_op_attention_split(sdm_ctx* ctx, const float* q, const float* k, const float* v, const float* bias, int B, int heads, int Lq, int Lk, float* out);
/* Antialiased bilinear resize of fp32 planes [P, Hin, Win] -> [P, Hout, Wout] (torchvision Resize). */
int sdm_op_resize_aa(sdm_ctx* ctx, const float* in, int P, int Hin, int Win, float* out, int Hout, int Wout);
/* Level-k additive key bias (natural-log domain) from the [-1,1] trimap plane [B,S,S] -> [B,(S/8>>k)^2]. */
int sdm_op_mask_bias(sdm_ctx* ctx, const float* trimap_plane, int B, int S, int level, float* bias_out);

#ifdef __cplusplus
}
#endif
#endif /* SDMATTE_H_ */
