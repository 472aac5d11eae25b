/* wqaa.h - C ABI of libwqaa_hip.so: the MI355X (gfx950) W_q A_a matmul backend.
 *
 * This is the drop-in boundary.  In microsoft/BitBLAS every (config, tuned hint) pair is JIT-compiled
 * into its own shared object that exports exactly two symbols
 *
 *     extern "C" void init();
 *     extern "C" void call(T_A* A, int8_t* B, [T_A* LUT], [T_A* Scale], [T_A* Zeros | int8_t* QZeros],
 *                          [T_A* Bias], T_out* C, [int m], hipStream_t stream);
 *
 * (reference: bitblas/builder/wrapper/base.py:9-19, bitblas/builder/wrapper/tl.py:104-120 and
 * :254-305 for the dynamic-m dispatcher) and is driven through ctypes by
 * `Operator._forward_from_prebuild_lib` (bitblas/ops/operator.py:458-463) and `Linear.forward`
 * (bitblas/module/__init__.py:267-289).
 *
 * Here nothing is JIT-compiled: ONE prebuilt library holds the static kernel set, and the per-config
 * `call` becomes `wqaa_matmul(desc, ...)` with the config passed as a plain struct.  Pointer order is
 * the reference's prim_func order (tirscript/matmul_dequantize_impl.py:465-478).  All pointers are
 * device pointers owned by the caller; the library never synchronises.  A launch runs on the device that owns
 * `stream` (made current for the duration of the call when it is not).  The split-K GEMM members need scratch for
 * their fp32 partial sums (wqaa_workspace_bytes): pass it with wqaa_matmul_opts (caller-owned, the reference's
 * model: general_matmul/__init__.py:29, 456-457, 482), or let the library keep one slab per (device, stream) -
 * never shared between streams, retired instead of freed when it has to grow (a captured hipGraph stays valid),
 * and refused with WQAA_ERR_LAUNCH when it would have to grow during stream capture.
 *
 * Plain C: no torch, no HIP types in signatures (hipStream_t is passed as void*).
 */
#ifndef WQAA_H_
#define WQAA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WQAA_ABI_VERSION 4   /* 2: wqaa_matmul_opts, wqaa_workspace_bytes (+ wqaa_matmul_group, new symbols only); 3: the 48-byte
                              * wqaa_epilogue (residual / norm), wqaa_matmul_gate_up, wqaa_matmul_chain - a caller built against this
                              * header probes wqaa_abi_version() >= 3 before it passes the long epilogue or calls them;
                              * 4: wqaa_matmul_chain always runs its launches and wants every item's C (the persistent chain
                              * member and the wqaa_debug_chain_* test aids are gone) */

/* element types of A / C / Scale / Bias */
enum wqaa_dtype {
  WQAA_F16 = 0,
  WQAA_BF16 = 1,
  WQAA_F32 = 2,
  WQAA_I8 = 3,
  WQAA_I32 = 4,
  WQAA_E4M3 = 5, /* OCP float8_e4m3fn */
  WQAA_E5M2 = 6,
  WQAA_I4 = 7    /* A only: two's-complement nibbles, two per byte, low nibble first (A is (M, K/2) int8):
                  * the reference's W_int4 / W_int2 x A_int4 path (general_matmul/__init__.py:373-378,
                  * tilelang/dequantize/matmul_dequantize_mma.py:512-790).  W NATIVE 4-bit = two's-complement
                  * nibbles; 2-bit weights are zero-extended (ibid. :742-749), whatever their signedness */
};

/* weight source formats: (format, bits) pairs of Matmul.BITBLAS_TRICK_DTYPE_MAP
 * (bitblas/ops/general_matmul/__init__.py:324-345) */
enum wqaa_wformat {
  WQAA_W_UINT = 0,   /* uint{1,2,4,8}                          */
  WQAA_W_INT = 1,    /* int{1,2,4,8}: stored as u, value u-2^(b-1) (int1: {0,-1}) */
  WQAA_W_NF = 2,     /* nf4: LUT[u]                             */
  WQAA_W_FP4 = 3,    /* "fp4_e2m1" as the reference decodes it  */
  WQAA_W_E4M3 = 4,   /* e4m3 bytes dequantised to A_dtype       */
  WQAA_W_E5M2 = 5,
  WQAA_W_NATIVE = 6  /* W_dtype == A_dtype (dense path)         */
};

enum wqaa_zeros_mode { WQAA_Z_NONE = 0, WQAA_Z_ORIGINAL = 1, WQAA_Z_RESCALE = 2, WQAA_Z_QUANTIZED = 3 };

/* byte layout of B, both are the reference's checkpoint layouts, shape (N, K*bits/8):
 *   PLAIN : general_compress order (bitblas/quantization/utils.py:54-70)
 *   LOP3  : PLAIN followed by the LOP3 interleave that `fast_decoding=True` applies
 *           (bitblas/ops/lop3_permutate/lop3_permutate_impl.py:12-132); target width 16 for
 *           A=float16, 8 for A=int8, 4 for A=int4 */
enum wqaa_layout { WQAA_LAYOUT_PLAIN = 0, WQAA_LAYOUT_LOP3 = 1 };

enum wqaa_status {
  WQAA_OK = 0,
  WQAA_ERR_BAD_DESC = 1,      /* malformed descriptor / null pointer */
  WQAA_ERR_UNSUPPORTED = 2,   /* no kernel for this (dtype, shape) combination */
  WQAA_ERR_LAUNCH = 3,        /* hipLaunch failed; see wqaa_last_error_string */
  WQAA_ERR_NO_DEVICE = 4
};

typedef struct wqaa_matmul_desc {
  int32_t struct_size;   /* = sizeof(wqaa_matmul_desc), ABI guard */
  int32_t N;             /* out features  */
  int32_t K;             /* in features   */
  int32_t a_dtype;       /* wqaa_dtype of A */
  int32_t w_format;      /* wqaa_wformat  */
  int32_t w_bits;        /* 1,2,4,8,16    */
  int32_t out_dtype;     /* wqaa_dtype of C */
  int32_t group_size;    /* -1 => K       */
  int32_t with_scaling;  /* Scale (N, K/g) in A_dtype */
  int32_t zeros_mode;    /* wqaa_zeros_mode; Zeros (N,K/g) A_dtype, or QZeros (K/g, N*bits/8) int8 (N*bits a multiple of 8: BAD_DESC otherwise) */
  int32_t with_bias;     /* Bias (N,) added after the cast to out_dtype */
  int32_t w_layout;      /* wqaa_layout   */
  int32_t strict_reference; /* 1: the reference's definition to the letter - dequantised weight rounded to A_dtype
                               per element (TE graph, matmul_dequantize_impl.py:391-459), e4m3->f16 bit trick
                               (0 -> 2^-7, quantization.py:169-176), "uint8" weights read through the signed storage
                               type; 0: members that skip the intermediate rounding (M <= 2 exact-product GEMV) and
                               decode e4m3 per IEEE may be taken - closer to the real-valued product.
                               NUMERICS CONTRACT (what a binder may rely on; C_ref = the reference's definition, i.e. the TE graph
                               evaluated with fp32 / int32 accumulation, oracle/wqaa_oracle.py; rms over the output):
                                 integer accumulators (A int8 / int4):          bit-exact;
                                 strict_reference = 1, float16:                 |C - C_ref| <= 1e-3 |C_ref| + 1e-3 rms(C_ref), any K
                                                                                (the per-element rounding of the definition itself);
                                 strict_reference = 0, float16, M >= 3:         the same bound (MFMA members: the same per-element decode);
                                 strict_reference = 0, float16, M <= 2 (exact-product GEMV: the dequantised weight is never rounded):
                                                                                the same 1e-3 + 1e-3 bound against the DEFINITION for
                                                                                K >= 4096 with group-wise scales and zeros none / original /
                                                                                quantized (measured <= 9.3e-4 rms over the Llama-sized shapes);
                                                                                1e-3 |C_ref| + 2e-3 rms(C_ref) where few products per output
                                                                                meet the definition's own per-element rounding, which these
                                                                                members skip: K < 4096, one group per row (per-channel), or
                                                                                `rescale` zero points (q s and z are rounded before they
                                                                                cancel) - measured worst cases 1.43e-3 rms on the reference's
                                                                                K = 256 fixtures, 1.6e-3 at K = 1024 with one group per row,
                                                                                1.4e-3 at K = 2112 rescale (profiles/r05_parity_margins.txt).
                                                                                Against the REAL-valued product these members are within
                                                                                1e-3 |C| + 1e-3 rms at any K (tests/test_gemvx_gpu.py,
                                                                                test_linear_gpu.py, tests/helpers.py: contract);
                                 bfloat16 outputs:                              8e-3 (the 2^-8 rounding of the result itself). */
  int32_t k_split_hint;  /* 0 / 1: the selector decides.  > 1: the caller's split-K request, as `MatmulConfigWithSplitK.k_split`
                            (ops/general_matmul_splitk.py:21-23).  Honoured where K is split by a free parameter: the
                            K split across the waves of a workgroup of the M <= 2 exact-product GEMV, and the split-K
                            count of the pipelined MFMA members (clamped to the k-steps available).  Members whose split is
                            structural (one-launch decode member: 8 waves; skinny member: 4 k-steps per workgroup) keep it;
                            wqaa_plan.split_k reports what was taken.  (was reserved[0], must-be-zero: ABI compatible) */
  int32_t two_pass_min_m; /* 0: never.  > 0: from this many activation rows on, run the TWO-PASS member where it exists -
                             B_decode written once to a scratch in A_dtype (the TE graph's first stage,
                             matmul_dequantize_impl.py:391-449, by the kernels' own decode routines), then the plain GEMM
                             through the vendor library - instead of the fused MFMA member.  Which of the two is faster is
                             shape-dependent (the library's heuristic), so this is a TUNED value: `Matmul.hardware_aware_
                             finetune` times both on the device, like the reference's tuner picks its hint
                             (ops/operator.py:262-293).  Needs N*K*sizeof(A_dtype) more scratch (wqaa_workspace_bytes).
                             (was reserved[0], must-be-zero: ABI compatible) */
  int32_t reserved[1];
} wqaa_matmul_desc;

/* what the selector chose for (desc, m): reported for tests, rocprof attribution and the cache */
typedef struct wqaa_plan {
  int32_t kernel_family;  /* 0 none, 1 gemv (VALU dot), 2 gemm (MFMA), 3 vendor-library GEMM (hipBLASLt: plain dense pairs, M >= 16), 4 two-pass (B_decode to scratch + 3) */
  int32_t block_m, block_n, block_k;
  int32_t threads;
  int32_t grid;
  int32_t rows_per_wave;  /* gemv */
  int32_t batch_tile;     /* gemv: activation rows handled per launch */
  int32_t pipeline_depth;
  int32_t split_k;
  int32_t lds_bytes;
  char name[96];          /* matmul_[m..]n..k.._AxW_<tile> style name
                             (general_matmul/__init__.py:240-318) */
} wqaa_plan;

/* ---- lifecycle ------------------------------------------------------------------------------
 * init(): idempotent; mirrors the generated `init()` (wrapper/base.py:5-13) - it raises the dynamic
 * LDS limit of every kernel that needs it.  Safe to call without a GPU (returns without touching
 * the device). */
void init(void);
int wqaa_abi_version(void);
int wqaa_device_count(void);

/* ---- the hot path ---------------------------------------------------------------------------
 * replaces the generated `call` (wrapper/tl.py:104-120, dynamic form :254-305).
 * A: (m, K) a_dtype row-major.  B: (N, K*bits/8) bytes.  C: (m, N) out_dtype.
 * LUT/Scale/Zeros/Bias may be NULL when the descriptor says they are absent.
 * m == 0 returns immediately (wrapper/tl.py:277).  Asynchronous on `stream`.
 * Returns wqaa_status; the reference's `call` is void, so the shim also records the error for
 * wqaa_last_error(). */
int wqaa_matmul(const wqaa_matmul_desc* desc, const void* A, const void* B, const void* LUT,
                const void* Scale, const void* Zeros, const void* Bias, void* C, int m, void* stream);

/* same launch through hipExtLaunchKernel with start/stop hipEvent_t.  Measured on MI355X / ROCm 7.2: the
 * event pair reports ~4.0 us for an EMPTY 512-workgroup kernel (rocprofv3 sees 1.4 us for the same kernel
 * launched plainly), so this is only meaningful for kernels of tens of microseconds; bench.py times
 * hipGraph replays instead (profiles/r01_floor_bench.txt). */
int wqaa_matmul_timed(const wqaa_matmul_desc* desc, const void* A, const void* B, const void* LUT,
                      const void* Scale, const void* Zeros, const void* Bias, void* C, int m,
                      void* stream, void* start_event, void* stop_event);

/* ---- callers' pre/post ops of the int8 path, fused at the boundary --
 * BitNet-style layers (integration/BitNet/utils_quant.py:161-216) wrap the W_int2 x A_int8 matmul in
 * two tiny torch.compile kernels: a per-token absmax quantiser before it and `out / si / sw -> half`
 * after it.  `wqaa_act_quant_int8` is the first; `wqaa_matmul_ex` folds the second into the matmul's
 * epilogue:  C[m, n] = half( (float(acc[m, n]) / row_scale[m]) / tensor_scale ) (+ half Bias[n]).
 * Only for int8 activations with out_dtype float16; desc.with_bias then means a float16 bias. */
#define WQAA_EPI_QUANTIZE_INPUT 1   /* A is the layer's float16 input (m, K): the kernel applies activation_quant
                                     * itself (per-token absmax -> int8) and uses its own si; row_scale is ignored
                                     * and may be NULL.  One launch for quantise + matmul + rescale; m <= 4 only
                                     * (WQAA_ERR_UNSUPPORTED otherwise: quantise with wqaa_act_quant_int8 first) */
/* ---- callers' elementwise ops of the float16 decode path (Llama-style MLP / residual stream) ---------------------
 * Between the projections of a decoder layer the reference's callers run elementwise kernels of a few KB each - the gated
 * activation `act_fn(gate_proj(x)) * up_proj(x)` (integration/BitNet/modeling_bitnet.py: BitnetMLP.forward :240-244 and
 * its fused gate/up twin :281-287) and the residual adds behind o_proj / down_proj (BitnetDecoderLayer.forward :839-860).
 * On MI355X every such launch is a ~1.3 us dependent boundary plus a tiny kernel next to a 4-8 us GEMV.  Both fold into the
 * GEMV that PRODUCES the vector, where each output element is in one lane's hands:
 *   WQAA_EPI_ADD_RESIDUAL  (wqaa_matmul_ex)  C[m, n] = half(float(out[m, n]) + float(residual[m, n])), out = the float16
 *                          result (bias included): torch's `residual + linear(x)`.  residual may alias C (each element
 *                          is read before it is written, by the same wave)
 *   wqaa_matmul_gate_up    (below)           act[m, n] = half(silu(gate_out[m, n])) * up_out[m, n]: gate_proj and up_proj
 *                          in ONE launch whose waves hold row n of both
 * and the norm in front of q/k/v and gate/up (BitnetRMSNorm = LlamaRMSNorm, modeling_bitnet.py:89-104; decoder layer
 * :841, :858) into the GEMV that CONSUMES the vector - a reduction over K, which every workgroup holds anyway:
 *   WQAA_EPI_RMSNORM_INPUT (wqaa_matmul_ex, wqaa_matmul_group_ex, wqaa_matmul_gate_up)  A is the hidden state x (m, K) float16;
 *                          the kernel stages norm_weight[k] * half(float(x[k]) * rsqrt(mean_k(x^2) + norm_eps)) - the
 *                          reference's two roundings; the fp32 sum of squares is taken in the kernel's own order (the
 *                          result is inside 1e-3 of torch's, not bit-identical).  Needs K <= 8192 at 4 bit (the rows
 *                          of x within the registers a workgroup loads ahead); not together with WQAA_EPI_ADD_RESIDUAL
 * For float16 activations x 1 / 2 / 4-bit integer weights, float16 output, m <= 2 (the exact-product GEMV members,
 * whatever desc.strict_reference says); WQAA_ERR_UNSUPPORTED otherwise - the caller then runs its own elementwise ops.
 * row_scale / tensor_scale are not used by WQAA_EPI_ADD_RESIDUAL. */
#define WQAA_EPI_ADD_RESIDUAL 2
#define WQAA_EPI_RMSNORM_INPUT 4
typedef struct wqaa_epilogue {
  int32_t struct_size;      /* = sizeof(wqaa_epilogue); the 24-byte prefix (up to reserved2) of earlier callers is accepted */
  int32_t flags;            /* 0 or an OR of WQAA_EPI_* */
  const float* row_scale;   /* (m,) si of activation_quant, device pointer */
  float tensor_scale;       /* sw = 1 / mean|W| */
  int32_t reserved2;
  const void* residual;     /* WQAA_EPI_ADD_RESIDUAL: (m, N) float16 */
  const void* norm_weight;  /* WQAA_EPI_RMSNORM_INPUT: (K,) float16 */
  float norm_eps;           /* WQAA_EPI_RMSNORM_INPUT: variance_epsilon */
  int32_t reserved3;
} wqaa_epilogue;

int wqaa_matmul_ex(const wqaa_matmul_desc* desc, const void* A, const void* B, const void* LUT,
                   const void* Scale, const void* Zeros, const void* Bias, void* C, int m, void* stream,
                   const wqaa_epilogue* epilogue);

/* ---- call options: caller-owned workspace and/or the fused epilogue above ------------------------
 * wqaa_workspace_bytes(desc, m): bytes of scratch the member selected for (desc, m) needs (0 for most members;
 * ksplit * m * N * 4 for the split-K GEMM members).  Needs no device.
 * wqaa_matmul_opts: as wqaa_matmul / wqaa_matmul_ex; `workspace` (16-byte aligned device memory of at least
 * wqaa_workspace_bytes, or NULL = library pool) must not be used by another stream at the same time. */
typedef struct wqaa_call_opts {
  int32_t struct_size;              /* = sizeof(wqaa_call_opts) */
  int32_t flags;                    /* reserved, 0 */
  void* workspace;
  uint64_t workspace_bytes;
  const wqaa_epilogue* epilogue;    /* NULL: plain matmul */
} wqaa_call_opts;

uint64_t wqaa_workspace_bytes(const wqaa_matmul_desc* desc, int m);
int wqaa_matmul_opts(const wqaa_matmul_desc* desc, const void* A, const void* B, const void* LUT,
                     const void* Scale, const void* Zeros, const void* Bias, void* C, int m, void* stream,
                     const wqaa_call_opts* opts);

/* ---- groups of independent operators: one launch for the q/k/v or gate/up projections of a layer ------------------
 * The reference has one `call` per operator; its own model integration fuses the projections of a decoder layer that
 * share an input by CONCATENATING their weights along N before quantisation (integration/BitNet/modeling_bitnet.py:
 * BitnetAttentionQKVFused.from_bit_attention :496-517, BitnetMLPFuseGateUp.from_bit_mlp :271-280, switched on by
 * default in quantize(fuse_qkv=True, fuse_gateup=True) :1433-1445).  This entry is the launch-level form of the same
 * fusion for operators that already exist as separate packed tensors (checkpoints with q/k/v and gate/up apart).
 * On MI355X a dependent kernel boundary costs ~1.3 us and a 4096 x 4096 int4 GEMV about 4 us, so at M <= 2 the boundaries are a third of a decoder layer.  wqaa_matmul_group runs `count` operators as
 * if wqaa_matmul had been called on each in turn - same results bit for bit, own pointers, own N - and, when they
 * share one GEMV tile configuration (same K, dtypes, format, group size and flags; M <= 2), in ONE launch whose grid
 * gives every member its own workgroups; no weight has to be re-packed or concatenated.  Groups the fused path does
 * not cover (M > 2, mixed configurations, more than WQAA_GROUP_MAX members) run as `count` launches in order.
 * Members must not alias each other's outputs (they run concurrently); inputs may be shared.
 * wqaa_group_plan reports the number of launches the group takes and, for a fused group, its plan (name suffix
 * "_x<count>"; tile configuration = the selector's choice for the merged operator, N = the sum of the members' rows). */
#define WQAA_GROUP_MAX 8
typedef struct wqaa_group_item {
  const wqaa_matmul_desc* desc;
  const void* A;
  const void* B;
  const void* LUT;
  const void* Scale;
  const void* Zeros;
  const void* Bias;
  void* C;
} wqaa_group_item;

int wqaa_matmul_group(const wqaa_group_item* items, int count, int m, void* stream);
/* as wqaa_matmul_group, every member with the fused epilogue of wqaa_matmul_ex (`epilogues[i]`, all members or NULL):
 * the q/k/v projections of a BitNet layer - in-kernel activation quantiser (WQAA_EPI_QUANTIZE_INPUT), W_int2 x A_int8,
 * `out / si / sw -> half` - as ONE launch (integration/BitNet/utils_quant.py:205-216 runs three such layers back to
 * back).  Members with different epilogue kinds run one by one. */
int wqaa_matmul_group_ex(const wqaa_group_item* items, const wqaa_epilogue* const* epilogues, int count, int m, void* stream);
int wqaa_group_plan(const wqaa_matmul_desc* const* descs, int count, int m, int* launches, wqaa_plan* plan);

/* gate_proj and up_proj of a gated MLP with the activation between them and down_proj folded in (see "callers' elementwise
 * ops" above): act[m, n] = half(silu(g)) * u with g, u = the float16 values the exact-product GEMV family stores for `gate` and
 * `up` - what wqaa_matmul gives wherever it takes that family itself (strict_reference = 0 at m = 1, and the m = 2 shapes its
 * selector keeps there), what wqaa_matmul_ex with a zero residual gives everywhere; a descriptor with strict_reference = 1 runs
 * the per-element-rounding members under wqaa_matmul and differs from g, u by that rounding (silu in fp32: g / (1 + exp(-g)), rounded to float16; then the float16 product - the roundings of torch's
 * `F.silu(gate) * up`).  One launch: every wave streams row n of BOTH weights against the shared input and the lane holding
 * the two sums stores one value - neither projection's output goes to memory.  gate->C / up->C are ignored (may be NULL);
 * gate->A == up->A; the two descriptors must agree in everything (N, K, format, group size, flags).
 * `norm`: NULL, or an epilogue with WQAA_EPI_RMSNORM_INPUT - A is then the hidden state in front of the MLP's norm.
 * wqaa_gate_up_plan: the plan of that launch (name suffix "_pair" / "_pair_norm") or WQAA_ERR_UNSUPPORTED, without a device. */
int wqaa_matmul_gate_up(const wqaa_group_item* gate, const wqaa_group_item* up, void* act, int m, void* stream,
                        const wqaa_epilogue* norm);
int wqaa_gate_up_plan(const wqaa_matmul_desc* desc, int m, int with_norm, wqaa_plan* plan);

/* ---- chains of DEPENDENT operators: the post-attention half of a decoder layer described once ------------------------------
 * o_proj (+ residual) -> RMSNorm -> gate / up * silu -> down_proj (+ residual) is four calls of this library whose only
 * coupling is a few KB of activations (the reference's callers: integration/BitNet/modeling_bitnet.py: BitnetMLP.forward
 * :240-244, BitnetDecoderLayer.forward :839-860, one `call` per nn.Linear through ops/operator.py:458-463).
 * wqaa_matmul_chain runs `count` items as the launches they stand for, in order -
 *     kind 0:  C = wqaa_matmul_ex(desc, input, B, Scale, Zeros, Bias; RMSNorm in front when norm_weight, residual added when
 *              residual / residual_from - never both on one item)
 *     kind 1:  C = wqaa_matmul_gate_up(gate = (B, Scale, Zeros, Bias), up = (B2, Scale2, Zeros2, Bias2); norm when norm_weight)
 * with `input` = A (input_from < 0) or the output C of the earlier item input_from, and the residual likewise.  Every item
 * stores its output: C must not be NULL.  (ABI 3 also carried a persistent one-launch member behind this entry point; it was
 * bit-identical to the launches and slower than them, and was removed in ABI 4 together with its wqaa_debug_chain_* aids.)
 * wqaa_chain_plan: validates the chain; *launches = count. */
#define WQAA_CHAIN_MAX 8
typedef struct wqaa_chain_item {
  const wqaa_matmul_desc* desc;
  const void* A;            /* (m, K) float16 when input_from < 0 */
  const void* B;
  const void* Scale;
  const void* Zeros;
  const void* Bias;
  const void* B2;           /* kind 1: the `up` operator's tensors (same descriptor) */
  const void* Scale2;
  const void* Zeros2;
  const void* Bias2;
  void* C;                  /* (m, N) float16 */
  const void* residual;     /* (m, N) float16 added to the rounded result (kind 0), or NULL */
  const void* norm_weight;  /* (K,) float16: RMSNorm in front of the operator, or NULL */
  float norm_eps;
  int32_t kind;             /* 0 matmul, 1 gate / up pair */
  int32_t input_from;       /* -1: A; i: output of item i (i < own index) */
  int32_t residual_from;    /* -1: `residual` (or none); i: output of item i */
} wqaa_chain_item;

int wqaa_matmul_chain(const wqaa_chain_item* items, int count, int m, void* stream);
int wqaa_chain_plan(const wqaa_chain_item* items, int count, int m, int* launches, wqaa_plan* plan);

/* measured tuning of the vendor-library GEMM behind (desc, m) - the plain dense pairs, or the second pass of the two-pass
 * member: the heuristic's top candidates are timed on the device (temporary buffers, synchronises `stream`) and the
 * fastest is kept for the process.  The counterpart of the reference's profiler pass (ops/operator.py:262-293); called by
 * `Matmul.hardware_aware_finetune`.  No-op without a device or where no library GEMM is involved. */
int wqaa_tune(const wqaa_matmul_desc* desc, int m, void* stream);

/* B_decode on its own: out (N, K) row-major in A_dtype = every weight decoded and (zero, scale)-dequantised as the TE
 * definition's first stage does (matmul_dequantize_impl.py:391-449) by the routines the MFMA members use in their loop.
 * float16 / bfloat16 / int8 activations' operators; K a multiple of 128 (256 for int8).  Asynchronous on `stream`. */
int wqaa_dequantize(const wqaa_matmul_desc* desc, const void* B, const void* LUT, const void* Scale, const void* Zeros,
                    void* out, void* stream);

/* per-row absmax quantiser (utils_quant.py:161-168): s = (1 / max(|x|, 1e-5)) * 127 - two fp32 roundings, what torch
 * evaluates for the reference's `Qp / tensor` (Tensor.__rtruediv__) -, q = clamp(rint(x * s)).
 * X: (rows, K) float16; Q: (rows, K) int8; S: (rows,) float32.  K % 8 == 0. */
int wqaa_act_quant_int8(const void* X, int64_t rows, int K, void* Q, float* S, void* stream);

/* tile-config selector: replaces roller + tuner (bitblas/base/roller, bitblas/base/tuner.py).  Needs no device.
 * The WQAA_GEMM_TUNE / WQAA_GEMV_TUNE tuning environment variables (csrc/wqaa_common.h: knob) are read here and at the first wqaa_matmul of a
 * (desc, m) pair per thread; a later change takes effect at the next wqaa_select call. */
int wqaa_select(const wqaa_matmul_desc* desc, int m, wqaa_plan* plan);
/* the member wqaa_matmul_ex takes for (desc, m) with an epilogue of these WQAA_EPI_* flags (0: the caller's row / tensor scales
 * alone - the output type and with it the tile choice differ from the plain call's; < 0: as wqaa_select) */
int wqaa_select_ex(const wqaa_matmul_desc* desc, int m, int epilogue_flags, wqaa_plan* plan);

/* ---- weight pre-processing (CPU; replaces the TVM-llvm ops of Matmul.transform_weight,
 * bitblas/ops/general_matmul/__init__.py:662-711: QuantCompress + LOP3Permutate) ---------------
 * codes: (rows, cols) int8 unsigned field values; out: (rows, cols*bits/8) bytes. */
int wqaa_pack_weight(const int8_t* codes, int64_t rows, int64_t cols, int bits, int layout,
                     int a_dtype, int8_t* out);
/* inverse: bytes in either layout -> unsigned field values */
int wqaa_unpack_weight(const int8_t* packed, int64_t rows, int64_t cols, int bits, int layout,
                       int a_dtype, int8_t* codes);
/* packed bytes of one layout -> packed bytes of the other, word by word (the reference's LOP3Permutate stage on its own:
 * ops/lop3_permutate/lop3_permutate_impl.py:12-132 is PLAIN -> LOP3; the inverse serves checkpoints that have to go back).
 * packed/out: (rows, row_bytes) bytes, row_bytes % 4 == 0; may not alias unless equal layouts.  All three functions split
 * the rows of large tensors over host threads (WQAA_PACK_THREADS overrides the count; 1 = serial). */
int wqaa_relayout_weight(const int8_t* packed, int64_t rows, int64_t row_bytes, int bits, int from_layout, int to_layout,
                         int a_dtype, int8_t* out);

/* ---- device self-test helper: decode `nwords` 32-bit words of packed weights with the kernels'
 * own decode routines (the HIP twin of testing/cpp/lop3_type_conversion/ *.cu known-answer tests).
 * out receives nwords*(32/bits) values: float16 (a_dtype F16) or int8 (a_dtype I8). */
int wqaa_debug_decode(const void* packed_dev, int64_t nwords, int w_format, int bits, int layout,
                      int a_dtype, int strict_reference, const void* lut_dev, void* out_dev,
                      void* stream);

/* host twin of the GEMV kernels' workgroup -> row-group-block map (csrc/wqaa_kinds.h xcd_row_blocks): workgroup `b` of a
 * grid of `grid` works on blocks out3[0], out3[0] + out3[1], ... < out3[2] of `n_blocks`.  Test aid, needs no device. */
void wqaa_debug_row_blocks(int b, int grid, int n_blocks, int* out3);
/* host twin of the MFMA members' workgroup -> tile map (k-slice, M-tile, N-tile; out4[3] = 1 when the division-free form is
 * in use) for a grid of tiles_m * tiles_n * ksplit workgroups: every tile of every k-slice is taken exactly once (test aid) */
void wqaa_debug_tile_of_block(int tiles_m, int tiles_n, int ksplit, int group_m, int block, int* out4);

/* ---- M = 1 output exchange of the column-parallel operator without a collective (bitblas_amd/parallel.py,
 * ColumnParallelMatmul(direct_store=True); the reference has no multi-GPU path: SURVEY.md section 5) ----------------------
 * A rank owns a WINDOW (wqaa_peer_alloc: uncached device memory; offset 0 holds `world` 32-bit flag words, the [1, N] output
 * rows follow), exports it (64-byte hipIpc handle, exchanged by the host over torch.distributed), opens its peers' windows
 * and, after the GEMV that wrote its [1, N/P] slice into its own window, calls wqaa_peer_exchange: ONE launch stores the
 * slice into every peer's row at this rank's columns, posts `step` into the peer's flag word for this rank (system-scope
 * release) and waits until every peer's post of `step` has arrived in this rank's flag words (bounded: on a timeout
 * `*status` = 1 + the late peer, the launch ends).  Steps count up by one per exchange; a row slot may be reused once the
 * exchange after next has been enqueued (the host alternates two slots).  Not for stream capture (step is an argument). */
#define WQAA_PEER_MAX 16
#define WQAA_PEER_HANDLE_BYTES 64
typedef struct wqaa_peer_exchange_desc {
  const void* src;                  /* this rank's slice inside its own window, 16-byte aligned */
  size_t bytes;                     /* of the slice; a multiple of 16 */
  int world, rank;
  uint32_t step;
  uint32_t timeout_ms;              /* 0: 2000 */
  void* dst[WQAA_PEER_MAX];         /* peer p's row at this rank's columns (mapped by wqaa_peer_open); [rank] unused */
  uint32_t* post[WQAA_PEER_MAX];    /* peer p's flag word for this rank (inside p's mapped window) */
  const uint32_t* flags;            /* own flag words [world] */
  uint32_t* status;                 /* own device word, zero before the first exchange */
} wqaa_peer_exchange_desc;
int wqaa_peer_alloc(size_t bytes, void** ptr);
int wqaa_peer_free(void* ptr);
int wqaa_peer_export(const void* ptr, void* handle64);
int wqaa_peer_open(const void* handle64, void** ptr);
int wqaa_peer_close(void* ptr);
int wqaa_peer_exchange(const wqaa_peer_exchange_desc* desc, void* stream);

/* ---- error side channel ---------------------------------------------------------------------- */
int wqaa_last_error(void);
const char* wqaa_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif /* WQAA_H_ */
