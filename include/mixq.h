/*
 * mixq.h -- C ABI of libmixq_mi355x.so, the MI355X (gfx950) MixQ W8A8O16 linear operator.
 *
 * Drop-in boundary for the hot path of Qcompiler/MixQ_Tensorrt_LLM.  Each entry point names the
 * reference interface it replaces (paths relative to the reference tree).  Plain pointers and sizes only:
 * no C++ types, no torch types.  All device pointers are HIP device pointers; every launch is asynchronous
 * on `stream` (a hipStream_t passed as void*); nothing here allocates or synchronises.
 *
 * Return convention: 0 = success, non-zero = MIXQ_E_* (the reference always returns 0 and ignores
 * CUTLASS/cuBLAS status -- TsinghuaMixQPlugin.cpp:402,752, kernel/i8gemm.cu:190-192).
 * No function throws across this boundary.
 */
#ifndef MIXQ_H_
#define MIXQ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#else
#include <stdbool.h>
#endif

#define MIXQ_API __attribute__((visibility("default")))

/* ---- error codes ------------------------------------------------------------------------------ */
enum {
    MIXQ_OK = 0,
    MIXQ_E_BADARG = 1,    /* null pointer, negative size, nbDims out of range */
    MIXQ_E_SHAPE = 2,     /* K/N not a multiple of 16 (the reference's CUTLASS 16-B alignment), O > 256 ... */
    MIXQ_E_ALIGN = 3,     /* a device pointer is not 16-byte aligned */
    MIXQ_E_HIP = 4,       /* a HIP launch failed; hipGetLastError() has the detail */
    MIXQ_E_WORKSPACE = 5, /* workspace pointer is null where one is needed */
    MIXQ_E_STALE = 6,     /* mixq_weight_image_verify: the bytes behind a registered weight pointer are not the registered weight's */
};

/* ---- tensor descriptor ------------------------------------------------------------------------ */
/* POD mirror of nvinfer1::PluginTensorDesc as TensorRT 10 lays it out (Dims64 {int32 nbDims; int64 d[8]},
 * DataType, TensorFormat, float scale) -- the type behind the `inputDesc/outputDesc` arguments of
 * MixQPlugin::enqueue (TsinghuaMixQPlugin.h:53-54).  Only dims are read (TsinghuaMixQPlugin.cpp:390-399). */
#define MIXQ_MAX_DIMS 8
typedef struct mixq_tensor_desc {
    int32_t nbDims;
    int64_t d[MIXQ_MAX_DIMS];
    int32_t type;   /* MIXQ_TYPE_HALF for all 8 tensors (supportsFormatCombination, .cpp:263-320) */
    int32_t format; /* MIXQ_FORMAT_LINEAR */
    float scale;
} mixq_tensor_desc;
enum { MIXQ_TYPE_FLOAT = 0, MIXQ_TYPE_HALF = 1, MIXQ_TYPE_INT8 = 2, MIXQ_TYPE_INT32 = 3 }; /* nvinfer1::DataType values */
enum { MIXQ_FORMAT_LINEAR = 0 };

/* nvinfer1::PluginField mirror for the creator path (TsinghuaMixQPlugin.cpp:895-933 reads "m","n","k" INT32). */
typedef struct mixq_plugin_field {
    const char* name;
    const void* data;
    int32_t type; /* MIXQ_FIELD_INT32 */
    int32_t length;
} mixq_plugin_field;
enum { MIXQ_FIELD_INT32 = 3 }; /* nvinfer1::PluginFieldType::kINT32 */

typedef struct mixq_handle mixq_handle; /* opaque; immutable after creation => enqueue is re-entrant */

/* ---- registry (MixQPlugins.cpp:33-132) -------------------------------------------------------- */
/* Same symbol, signature and behaviour the reference's loaders call (plugin.py:34-43, summarize.py:45-56,
 * run.py:41-52): registers creator ("MixQ","1",libNamespace) once per namespace, mutex-guarded, idempotent,
 * returns true.  `logger` is ignored (there is no TensorRT on MI355X). */
MIXQ_API bool initOpenAiTritonPlugins(void* logger, const char* libNamespace);
/* trt.get_plugin_registry().get_plugin_creator(name, version, ns) (plugin.py:55-57): 1 if registered. */
MIXQ_API int mixq_registry_has_creator(const char* name, const char* version, const char* libNamespace);
MIXQ_API const char* mixq_plugin_type(void);    /* "MixQ"  getPluginType  .cpp:775-778 */
MIXQ_API const char* mixq_plugin_version(void); /* "1"     getPluginVersion .cpp:780-783 */

/* ---- plugin object lifecycle (TsinghuaMixQPlugin.h:34-89) ------------------------------------- */
MIXQ_API mixq_handle* mixq_create(int32_t m, int32_t n, int32_t k);                       /* ctor .cpp:217-225 */
MIXQ_API mixq_handle* mixq_create_from_fields(const mixq_plugin_field* fields, int32_t nbFields); /* createPlugin .cpp:895-933 */
/* MixQPluginCreator::getFieldNames (.cpp:890-893, .h:100): the table the creator ADVERTISES, exactly as the reference fills it in its
 * constructor (.cpp:868-878): three INT32 fields named "mm", "mn", "mk", data == NULL, length == -1.  (createPlugin above parses
 * "m", "n", "k" -- .cpp:906-919 -- as the reference does; a host that enumerates the advertised names finds what the reference
 * would show it.)  Returns a pointer to a static array, *nbFields receives its length (3); never NULL. */
MIXQ_API const mixq_plugin_field* mixq_get_field_names(int32_t* nbFields);
MIXQ_API mixq_handle* mixq_deserialize(const void* data, size_t length);                  /* .cpp:227-234, 935-951 */
MIXQ_API size_t mixq_serialization_size(const mixq_handle* h);                            /* == 12, .cpp:808-811 */
MIXQ_API void mixq_serialize(const mixq_handle* h, void* buffer);                         /* .cpp:813-820 */
MIXQ_API mixq_handle* mixq_clone(const mixq_handle* h);                                   /* .cpp:237-242 */
MIXQ_API void mixq_destroy(mixq_handle* h);                                               /* .cpp:851-855 */
MIXQ_API int mixq_initialize(mixq_handle* h);                                             /* .cpp:792-799 (no cuBLAS handle to make) */
MIXQ_API void mixq_terminate(mixq_handle* h);                                             /* .cpp:801-806 */
MIXQ_API int mixq_get_mnk(const mixq_handle* h, int32_t* m, int32_t* n, int32_t* k);
MIXQ_API int mixq_set_namespace(mixq_handle* h, const char* ns);                          /* .cpp:857-860 */
MIXQ_API const char* mixq_get_namespace(const mixq_handle* h);                            /* .cpp:862-865 */

/* ---- shape / format negotiation --------------------------------------------------------------- */
MIXQ_API int mixq_get_nb_outputs(const mixq_handle* h); /* 1, .cpp:785-788 */
/* getOutputDimensions (.cpp:244-261): out = in0.dims with last dim replaced by in1.d[0]. */
MIXQ_API int mixq_get_output_dimensions(const mixq_handle* h, int outputIndex, const mixq_tensor_desc* inputs,
                                        int nbInputs, mixq_tensor_desc* out);
/* supportsFormatCombination (.cpp:263-320): pos 0..7 must be HALF + LINEAR. 1 = supported. */
MIXQ_API int mixq_supports_format_combination(const mixq_handle* h, int pos, const mixq_tensor_desc* inOut,
                                              int nbInputs, int nbOutputs);
MIXQ_API int mixq_get_output_data_type(const mixq_handle* h, int index); /* MIXQ_TYPE_HALF, .cpp:768-773 */

/* ---- workspace -------------------------------------------------------------------------------- */
/* configurePlugin + getWorkspaceSize (.cpp:325-378).  Bytes this implementation needs for M <= maxM:
 *   qA int8 [maxM*K] | sA fp16 [maxM] | fpA fp16 [maxM*128], each carved at 128-B alignment like
 *   nextWorkspacePtr (.cpp:206-215).  size_t-clean (the reference overflows int). */
MIXQ_API size_t mixq_workspace_size(const mixq_handle* h, int64_t maxM, int64_t N, int64_t K);
/* Exchange scratch that ONE mixq_enqueue call with exactly M rows carves behind fpA (K splits over workgroups; 0 for
 * most shapes).  Not monotone in M; mixq_workspace_size(maxM, N, K) covers the maximum over every M <= maxM (a caller
 * that sizes once, like TensorRT's getWorkspaceSize, is safe for every later M). */
MIXQ_API size_t mixq_enqueue_scratch_size(int64_t M, int64_t N, int64_t K);
/* The reference's own (larger) formula, for callers that size buffers by it:
 *   max(maxM*K + 2*maxM + 2*K*N, 16*maxM*N), 32 MiB if that is 0. */
MIXQ_API size_t mixq_reference_workspace_size(int64_t maxM, int64_t N, int64_t K);

/* ---- enqueue: THE hot path (MixQPlugin::enqueue, TsinghuaMixQPlugin.h:53-54, .cpp:384-765) ------
 * inputs[0] A fp16 [...,K] | [1] weight = int8 [N,K] (declared fp16 [N,K/2]) | [2] weights_scaling_factor fp16 [N]
 * | [3] fp_weight fp16 [N,128] | [4] fp_ind = int32 [128] (declared fp16 [256]) | [5] qweight = uint8 [K,N]
 * EETQ-interleaved (declared fp16 [K,N/2]) | [6] scaling_factors fp16 [N] ; outputs[0] fp16 [...,N].
 * M = prod(in0.d[:-1]), K = in0.d[-1], N = in1.d[0].   M > 4: prefill path; M <= 4: W8A16 decode path.
 * Asynchronous on `stream`. */
MIXQ_API int mixq_enqueue(const mixq_handle* h, const mixq_tensor_desc* inputDesc, const mixq_tensor_desc* outputDesc,
                          const void* const* inputs, void* const* outputs, void* workspace, void* stream);

/* Same work, same launches; additionally records two caller-owned hipEvent_t (may be NULL) on `stream` immediately
 * before and after the fused-GEMM launch of the prefill path, so a harness can time the dominant kernel inside its
 * timed region without changing the path (bench.py "roofline"). */
MIXQ_API int mixq_enqueue_profiled(const mixq_handle* h, const mixq_tensor_desc* inputDesc,
                                   const mixq_tensor_desc* outputDesc, const void* const* inputs, void* const* outputs,
                                   void* workspace, void* stream, void* ev_gemm_start, void* ev_gemm_stop);

/* ---- launcher level (kernel/int8FusedDequantizeCUDA.h:3-27, weightonlykernel/fpA_intB_gemm_wrapper.h:8-11) ----- */
/* int8quant (kernel/i8gemm.cu:139-150): per-row fp16 amax/127 scale + int8 quantisation. */
MIXQ_API int mixq_int8quant(int rows, int cols, const void* src_f16, int8_t* output, void* scale_f16, void* stream);
/* ExtractOutliersAndSetToZeros (kernel/i8gemm.cu:226-244): fp_A[m,j] = A[m,ind[j]]; A is NOT modified (T-flavour). */
MIXQ_API int mixq_extract_outliers(int M, int K, const void* A_f16, void* fpA_f16, const int32_t* ind, int len,
                                   void* stream);
/* mixlib twin (quantkernel/mix_cuda/cult.cu:1406-1465): same gather, then writes 0 into A (P-flavour). */
MIXQ_API int mixq_extract_outliers_set_zero(int M, int K, void* A_f16, void* fpA_f16, const int32_t* ind, int len,
                                            void* stream);
/* The fused producer this implementation actually runs in enqueue: one pass over A producing qA, sA and fpA.
 * zero_outliers = 0: T-flavour (amax includes outlier columns, A untouched; == mixq_extract_outliers + mixq_int8quant).
 * zero_outliers = 1: mixlib FindRowScaleFusedExtracOutliers (cult.cu:2616-2709): outlier columns count as 0 for amax
 * and q, and A is written back with zeros in those columns. */
MIXQ_API int mixq_quant_extract(int M, int K, void* A_f16, int8_t* qA, void* sA_f16, void* fpA_f16, const int32_t* ind,
                                int len, int zero_outliers, void* stream);
/* mixlib Int8quantize (quantkernel/mix_cuda/cult.cu:1732-1771): output = (int8) half2int_rn(hdiv(src, scale[row])) with a
 * caller-supplied per-row fp16 scale. */
MIXQ_API int mixq_int8_quantize_with_scale(int rows, int cols, const void* src_f16, const void* scale_f16,
                                           int8_t* output, void* stream);
/* SURVEY §8f row 1 -- the producer in front of the operator (P-flavour fused norm):
 * layernorm_forward_cuda (quantkernel/mix_cuda/layernorm/layernorm.cu:100-117): out = fp16(clamp((x*rstd)*gamma)), T5-style RMSNorm. */
MIXQ_API int mixq_rmsnorm(int M, int K, const void* x_f16, const void* gamma_f16, void* out_f16, float eps, void* stream);
/* layernorm_forward_cuda_extract_outliers (layernorm.cu:122-198, 316-346): RMSNorm, then outliers[m,j] = out[m,ind[j]],
 * those columns zeroed in `out`, then per-row scale + int8 quantisation of the zeroed row -- one pass over x. */
MIXQ_API int mixq_rmsnorm_extract_quant(int M, int K, const void* x_f16, const void* gamma_f16, void* out_f16, float eps,
                                        const int32_t* ind, int len, void* outliers_f16, int8_t* q, void* scale_f16,
                                        void* stream);
/* int8FusedDequantizeCUDA (kernel/i8gemm.cu:151-194): D = fp16(float(A.B^T) * (scale_col[n]*scale_row[m]) + y).
 * A int8 [M,K], B int8 [N,K], scale_row fp16 [M], scale_col fp16 [N], y/D fp16 [M,N] (y may alias D, may be NULL = 0).
 * `workspace` (the reference hands CUTLASS its scratch here): NULL, or device scratch for the K split over workgroups --
 * at least mixq_gemm_scratch_size(M,N,K) bytes, ZERO-FILLED before its first use, one per stream (see
 * mixq_gemm_mixed_scratch below).  NULL always selects the one-workgroup-per-tile kernels. */
MIXQ_API int mixq_int8_fused_dequantize(const int8_t* A, const int8_t* B, const void* scale_row, const void* scale_col,
                                        const void* y, void* D, int M, int N, int K, char* workspace, void* stream);
/* Same with the SiLU epilogue (mixlib int8FusedDequantizeSilu, linear_combination_dequant.h:176-270). */
MIXQ_API int mixq_int8_fused_dequantize_silu(const int8_t* A, const int8_t* B, const void* scale_row,
                                             const void* scale_col, const void* y, void* D, int M, int N, int K,
                                             char* workspace, void* stream);
/* The single fused GEMM this implementation runs in enqueue: int8 main loop + fp16 outlier side-GEMM (O columns,
 * fp32 accumulate, rounded to fp16 like the reference's separate cuBLAS call) + dequant epilogue, Out written once. */
MIXQ_API int mixq_gemm_mixed(const int8_t* qA, const int8_t* W, const void* sA, const void* sW, const void* fpA,
                             const void* fpW, void* Out, int M, int N, int K, int O, void* stream);
/* The fused GEMM with caller-owned device scratch (MI355X extension).  Mid-size problems whose 256x256 tiles cover at
 * most 1/2, 1/4, 1/8 of the CUs split K over 2 / 4 / 8 workgroups per tile, which exchange their int32 partial sums
 * through `scratch` and each finish a share of the tile: bit-identical results, 10-35 % less time on the shapes it is
 * chosen for (csrc/gemm_pp_kernels.hip).  mixq_gemm_scratch_size(M,N,K) = bytes needed (0: the split
 * form is not used for this shape; never more than ~56 MiB).  The scratch must be ZERO-FILLED before its first use (its
 * first 16 KiB hold the hand-over words of every shape; each launch leaves them zero again, so one scratch serves
 * launches of any shapes in stream order) and must not be shared by launches that can run concurrently (one scratch
 * per stream).  A null / too small scratch selects the one-workgroup-per-tile kernels (= mixq_gemm_mixed).
 * Decode batches / short chunks on narrow outputs with K >= 8192 use the same scratch for the small-tile kernels' K split
 * (2..16 workgroups per 32x64 / 64x64 tile, "last block finishes": csrc/gemm_kernels.hip).
 * mixq_enqueue carves this scratch from the plugin workspace itself; its quantiser clears the hand-over words on every
 * call (no extra launch).  mixq_debug_set_gemm_variant(60 / 62 / 64 / 68 / 66 / 69) = small-tile split off / 2 / 4 / 8 /
 * 16 ways / automatic.
 * mixq_debug_set_gemm_variant(70) switches the split form off, 72 / 74 / 78 force a factor, 79 = automatic (default). */
MIXQ_API size_t mixq_gemm_scratch_size(int M, int N, int K);
/* Upper bound of mixq_gemm_scratch_size over all shapes on the current device (~56 MiB on MI355X): allocate the
 * per-stream scratch ONCE at this size and its address never changes -- which a captured HIP graph relies on. */
MIXQ_API size_t mixq_gemm_scratch_bound(void);
MIXQ_API int mixq_gemm_mixed_scratch(const int8_t* qA, const int8_t* W, const void* sA, const void* sW, const void* fpA,
                                     const void* fpW, void* Out, int M, int N, int K, int O, void* scratch,
                                     size_t scratch_bytes, void* stream);
/* ---- weight images (MI355X extension; no reference counterpart, no result changes) -----------------------------------------
 * The reference stores `weight` (inputs[1]) row-major int8 [N, K].  Decode batches (5 .. 64 rows) are weight streams, and the
 * weight-streaming GEMM's MFMA fragment load takes 64 bytes of 16 rows that lie K bytes apart.  A runtime that can spare N * K
 * bytes per layer registers a FRAGMENT-MAJOR copy once at load time; from then on every call on that `weight` POINTER whose shape
 * the decode-batch GEMM serves -- mixq_enqueue, mixq_gemm_mixed*, mixq_int8_fused_dequantize*, mixq_mixlinear_forward -- reads
 * the copy (one contiguous 1-KiB read per load; images of 32 MiB and more with non-temporal loads).  Bit-identical results;
 * operator time at 32 rows -10..-15 % (profiles/r04_weight_image_probe.txt).  Every other kernel keeps reading `weight` itself,
 * which must stay valid and unchanged.
 *   mixq_weight_image_bytes(N, K)        bytes the image needs (N * K), or 0 if the shape has no image (N % 16 or K % 64 != 0)
 *   mixq_weight_image_register(...)      builds the image of `weight` into caller-owned device memory `image` and registers it under the
 *                                        pointer `weight` together with a 64-bit CONTENT TAG of the weight; registering again replaces it.
 *                                        Set-up work: two passes over the weight on `stream`, which it SYNCHRONISES (the only entries of
 *                                        this header that do, with _verify)
 *   mixq_weight_image_unregister(weight) forgets it -- REQUIRED before `weight` or `image` is freed or rewritten
 *   mixq_weight_image_verify(weight, s)  MIXQ_OK if the bytes behind `weight` still carry the registered tag, MIXQ_E_STALE if not (the
 *                                        entry is dropped: calls go back to reading `weight`), MIXQ_E_BADARG if nothing is registered (or the
 *                                        check itself failed: the entry is dropped as well); synchronises `s`
 * An address says nothing about what lives there: a weight freed and re-allocated at the same address with the same shape must not be
 * served the old tensor's image.  The image of a registration is therefore NOT TRUSTED until the bytes behind the pointer have been
 * compared with the tag once more around its FIRST USE -- without blocking anybody: the first call on a registered weight enqueues the
 * tag kernel (one pass over the weight) and an 8-byte copy on its own stream and reads `weight` itself, like every call until that work
 * has completed; the first call that finds it complete compares the tags (equal: the image is streamed from then on; different: the
 * entry is dropped and counted in mixq_weight_image_stale_count()).  The hot entries never allocate, free or synchronise (round 6:
 * everything the check needs is allocated by _register).  A first use inside a stream capture cannot enqueue the check and reads `weight`
 * itself -- call mixq_weight_image_verify after registering, or run a call eagerly before capturing, as any graph warm-up does.  After
 * the check the contract is the usual one for a pointer held by a library: unregister before the memory changes hands (a captured graph
 * holds the image pointer like any other argument: keep image and registration alive as long as the graph).
 * Process-global, thread-safe (readers share a lock; with nothing registered a lookup is one atomic load); every call looks its
 * weight up ONCE. */
MIXQ_API size_t mixq_weight_image_bytes(int64_t N, int64_t K);
MIXQ_API int mixq_weight_image_register(const int8_t* weight, int64_t N, int64_t K, void* image, void* stream);
MIXQ_API int mixq_weight_image_unregister(const int8_t* weight);
MIXQ_API int mixq_weight_image_verify(const int8_t* weight, void* stream);
MIXQ_API int mixq_weight_image_stale_count(void);

/* ---- qA layouts (MI355X extension; decode batches) --------------------------------------------------------------------------
 * The int8 activation image between a producer (quantiser, fused norm) and the fused GEMM is the library's own intermediate,
 * so the producer may write the layout the consuming kernel reads fastest.  ROW_MAJOR [M,K] is what every reference-named
 * entry above produces and consumes.  FRAGMENT_MAJOR (5..64 rows, K in (1024, 16384], shapes the weight-streaming skinny GEMM
 * serves -- mixq_qa_layout(M, N, K) decides): for 16-row tile t and 64-byte k-step s, the 1-KiB block t * ceil(K / 64) + s holds,
 * at lane * 16, the 16 bytes (row t * 16 + lane % 16, k = s * 64 + (lane / 16) * 16 ...) -- each qA load of that kernel becomes one
 * contiguous 1-KiB read instead of 16 rows x 64 bytes 4 KiB apart: 32 x 4096 x 4096 GEMM 7.9 -> 5.4 us (BASELINE configs[0]
 * 11.8 -> 9.2 us through mixq_enqueue, which uses it internally).  The image is opaque (mixq_qa_bytes bytes, 16-byte aligned)
 * and has ONE reader: passing it for a shape the skinny kernel does not serve returns MIXQ_E_SHAPE.  Same bits as row-major. */
#define MIXQ_QA_ROW_MAJOR 0
#define MIXQ_QA_FRAGMENT_MAJOR 1
MIXQ_API int mixq_qa_layout(int M, int N, int K);               /* the layout to produce for a [M,K] x [N,K]^T consumer without scratch */
MIXQ_API size_t mixq_qa_bytes(int M, int K, int layout);        /* bytes of the image */
MIXQ_API int mixq_quant_extract_layout(int M, int K, void* A_f16, int8_t* qA, void* sA_f16, void* fpA_f16, const int32_t* ind,
                                       int len, int zero_outliers, int q_layout, void* stream);
MIXQ_API int mixq_rmsnorm_extract_quant_layout(int M, int K, const void* x_f16, const void* gamma_f16, void* out_f16, float eps,
                                               const int32_t* ind, int len, void* outliers_f16, int8_t* q, void* scale_f16,
                                               int q_layout, void* stream);
/* int8FusedDequantize / ...Silu / ...SiluMul (epilogue 0 / 1 / 3) on a qA image of the given layout; mul NULL unless epilogue 3 */
MIXQ_API int mixq_int8_fused_dequantize_layout(const int8_t* A, const int8_t* B, const void* scale_row, const void* scale_col,
                                               const void* y, const void* mul, void* D, int M, int N, int K, int epilogue,
                                               int qa_layout, char* workspace, void* stream);
MIXQ_API int mixq_gemm_mixed_layout(const int8_t* qA, const int8_t* W, const void* sA, const void* sW, const void* fpA,
                                    const void* fpW, void* Out, int M, int N, int K, int O, int qa_layout, void* scratch,
                                    size_t scratch_bytes, void* stream);
/* MixLinear_GEMM.forward (MixQ/src/mixquant/modules/linear.py:163-286, bit = 8, static outlier set) as ONE call and TWO
 * launches: the reference makes four mixlib calls per linear here (ExtractOutliersAndSetToZeros, FindRowScale, torch.mm on the
 * outliers, int8FusedDequantize); a decode step is host-bound on those calls long before it is GPU-bound
 * (profiles/r03_mixlib_overhead.txt).  x fp16 [M,K] (its `ind` columns are ZEROED, as the reference does), ind int32 [O],
 * q_weight int8 [N,K], scale_col fp16 [N], weight_cache fp16 [N,O]; outputs: x_scale fp16 [M], q_x int8 [M,K], outliers fp16
 * [M,O], out fp16 [M,N].  q_layout: MIXQ_QA_ROW_MAJOR, or -- only where mixq_qa_layout(M, N, K) says so -- FRAGMENT_MAJOR (q_x then
 * is the opaque image of mixq_qa_bytes(M, K, 1) bytes).  scratch: as mixq_gemm_mixed_scratch (may be NULL). */
MIXQ_API int mixq_mixlinear_forward(int M, int N, int K, int O, void* x_f16, const int32_t* ind, const int8_t* q_weight,
                                    const void* scale_col, const void* weight_cache, void* x_scale, int8_t* q_x,
                                    void* outliers_f16, void* out_f16, int q_layout, void* scratch, size_t scratch_bytes,
                                    void* stream);
/* gemm (TsinghuaMixQPlugin.cpp:36-77, cuBLAS s8 x s8 -> s32): raw int32 accumulators, for bit-exact checks. */
MIXQ_API int mixq_gemm_s8s8s32(const int8_t* A, const int8_t* B, int32_t* C, int M, int N, int K, void* stream);
/* gemmfp16 (TsinghuaMixQPlugin.cpp:122-161): Out = fpA . fpW^T, fp32 accumulate, fp16 out. */
MIXQ_API int mixq_gemm_fp16(const void* fpA, const void* fpW, void* Out, int M, int N, int O, void* stream);
/* dequantizationCUDA (kernel/i8gemm.cu:258-300): out = hadd(fp16((float(x)*sRow[m])*sCol[n]), out). */
MIXQ_API int mixq_dequantization(void* out_f16, const int32_t* x, const void* scaleRow, const void* scaleCol, int M,
                                 int N, void* stream);
/* dequantizationCUDASilu (quantkernel/mix_cuda/cult.cu:2305-2348, mixlib dequantizeInt8Silu, the P-flavour's sm90 route
 * MixQ/src/mixquant/modules/linear.py:321-324): out = fp16( silu( (float(x)*sRow[m])*sCol[n] + float(y) ) ), fp32 math,
 * one rounding.  out may alias y. */
MIXQ_API int mixq_dequantization_silu(void* out_f16, const int32_t* x, const void* scaleRow, const void* scaleCol,
                                      const void* y_f16, int M, int N, void* stream);
/* w8_a16_gemm_forward_cuda (weightonlykernel/fpA_intB_gemm_wrapper.cu:29-70): Out = A . dequant(qweight), fp16 A [m,k],
 * weight = EETQ-interleaved uint8 [K,N] consumed as stored, scale fp16 [N], Out fp16 [m,n].
 * m <= 4: the batched GEMV (weightOnlyBatchedGemv, :53-58); m > 4: the fpA_intB tensor-core GEMM (ft::gemm_fp16_int,
 * :59-69 -> fpA_intB_gemm_template.h:441-552) -- here fp16 MFMA with the int8 weights dequantised in registers
 * (csrc/w8a16_gemm_kernels.hip).  k % 64 == 0, n % 2 == 0.  The reference passes its GEMM no workspace (nullptr, 0). */
MIXQ_API int mixq_w8a16_gemm_forward(const void* input_f16, const uint8_t* weight, const void* scale_f16,
                                     void* output_f16, int m, int n, int k, void* stream);
/* The same with caller-owned scratch (MI355X extension; CUTLASS' split-k workspace, fpA_intB_gemm_wrapper.cu:16-25, is the
 * counterpart): with few column tiles K is split over several workgroups per tile, which keeps every CU streaming
 * weights; large problems (from ~1300 tokens) run in two passes -- W dequantised once to fp16 in the scratch (n k 2 bytes
 * behind the hand-over words), then the fp16 ping-pong GEMM -- when the scratch is large enough, the fused form otherwise.
 * mixq_w8a16_gemm_workspace_size(m,n,k) bytes (0: nothing to gain for this shape), ZERO-FILLED before its first use,
 * one per stream; it may be the same buffer as mixq_gemm_mixed_scratch's (every kernel leaves the hand-over words zero).
 * Results do not depend on whether / how K was split in the last bit only up to fp32 summation order: the parts are
 * always added in the same order, so a given (shape, scratch-or-not) is deterministic. */
MIXQ_API size_t mixq_w8a16_gemm_workspace_size(int m, int n, int k);
MIXQ_API int mixq_w8a16_gemm_forward_ws(const void* input_f16, const uint8_t* weight, const void* scale_f16,
                                        void* output_f16, int m, int n, int k, void* workspace, size_t workspace_bytes,
                                        void* stream);

/* MLP fusion beyond the reference (MixQ/src/mixquant/modules/fused/mlp.py:57-64 runs int8FusedDequantizeSilu and then
 * `gate_output *= up_output` as a separate pass over [M,N]): D = fp16( fp16(silu(float(A.B^T)*(sc*sr) + y)) * mul ),
 * bit-identical to the two-step sequence, one pass.  mul = fp16 [M,N] (the up projection's output), 8-byte aligned. */
MIXQ_API int mixq_int8_fused_dequantize_silu_mul(const int8_t* A, const int8_t* B, const void* scale_row,
                                                 const void* scale_col, const void* y, const void* mul, void* D, int M,
                                                 int N, int K, char* workspace, void* stream);

/* ---- dynamic outliers of the P-flavour forward (MixQ/src/mixquant/modules/linear.py) ------------- */
/* FindOutliers (linear.py:155-161): torch.unique(torch.where(A.abs() > sigma)[1]) -> ascending int32 column indices of
 * fp16 A [M,K] that hold at least one |a| > sigma (NaN compares false).  `mask_ws` = device scratch of
 * mixq_find_outliers_workspace_size(K) bytes; at most `capacity` indices are written to ind_out, *count_out (device
 * int32) receives the true count.  Asynchronous on `stream`; K % 8 == 0, A 16-byte aligned. */
MIXQ_API size_t mixq_find_outliers_workspace_size(int K);
MIXQ_API int mixq_find_outliers(const void* A_f16, int M, int K, float sigma, void* mask_ws, int32_t* ind_out,
                                int32_t* count_out, int capacity, void* stream);
/* weight_cache columns (linear.py:207-209): out[n,j] = fp16( fp16(q_weight[n, ind[j]]) * scale_col[n] ), fp16 [N,len]. */
MIXQ_API int mixq_dequant_weight_columns(const int8_t* q_weight, const void* scale_col_f16, const int32_t* ind, int len,
                                         void* out_f16, int N, int K, void* stream);

/* ---- 4-bit (W4A4) flavour: the `bit == 4` branch of MixQ/src/mixquant/modules/linear.py -------------- */
/* Packed int4 = cutlass::int4b_t pairs in a byte: element 2i in the low nibble, 2i+1 in the high nibble.
 * FindRowScale(bit = 4) (cult.cu:2515-2567, 2588-2606): scale[m] = fp16(amax / 7), dst[m, i] = int4(rn(x/scale)) pairs;
 * src fp16 [rows, cols], dst uint8 [rows, cols/2].  cols % 8 == 0. */
MIXQ_API int mixq_int4quant(int rows, int cols, const void* src_f16, uint8_t* dst_packed, void* scale_f16,
                            void* stream);
/* int4FusedDequantize[Silu] (cult.cu:2005-2060, 2119-2181): D = fp16(float(A.B^T) * (scale_col[n]*scale_row[m]) + y)
 * with A uint8 [M, k_packed], B uint8 [N, k_packed] packed int4, K = 2 * k_packed (the reference also takes the packed
 * count).  gfx950 has no int4 MFMA; the int32 accumulators are nevertheless the reference's, bit for bit:
 *   M <= 64 (decode batches): ONE launch that streams the PACKED weight -- N * k_packed bytes, the HBM saving 4-bit weights are
 *     for -- and widens the nibbles in registers on their way into the int8 MFMA (csrc/int4_gemm_kernels.hip); `workspace` is not
 *     used and may be NULL (K < 131040 elements: beyond, the unpack route below);
 *   M > 64: both operands are sign-extended to int8 in `workspace` (mixq_int4_fused_workspace_size bytes, caller-owned, 16-byte
 *     aligned) and the int8 MFMA kernels run on them.  A caller that can spare N * K bytes per layer widens the weight ONCE at
 *     load time (mixq_unpack_int4_to_int8) and calls mixq_int4_fused_dequantize_w8 instead: per call only A is widened.
 * k_packed % 16 == 0, N % 16 == 0.  y may be NULL (no addend).
 * mixq_int4_fused_workspace_size(M, N, k_packed) = the bytes the call with these sizes REALLY needs: 0 where it streams the packed weight,
 * both widened operands otherwise; N = 0 asks for the activation's part alone (the _w8 entry below). */
MIXQ_API size_t mixq_int4_fused_workspace_size(int M, int N, int k_packed);
MIXQ_API int mixq_int4_fused_dequantize(const uint8_t* A, const uint8_t* B, const void* scale_row, const void* scale_col,
                                        const void* y, void* D, int M, int N, int k_packed, char* workspace,
                                        void* stream);
MIXQ_API int mixq_int4_fused_dequantize_silu(const uint8_t* A, const uint8_t* B, const void* scale_row,
                                             const void* scale_col, const void* y, void* D, int M, int N, int k_packed,
                                             char* workspace, void* stream);
/* MI355X extension (round 6): the 4-bit flavour's linear on an fp16 activation in ONE call -- FindRowScale(bit = 4) (cult.cu:2515-2567:
 * x_scale[m] = fp16(amax / 7), q = int4(rn(x / x_scale))) followed by int4FusedDequantize[Silu] above.  x fp16 [M, 2 * k_packed]; x_scale
 * fp16 [M] is WRITTEN (the P-flavour's cache.x_scale); epilogue 0 = dequant, 1 = dequant + SiLU.  ONE row runs as ONE launch: every
 * workgroup of the weight-streaming kernel quantises the row for itself and keeps the packed row in LDS (no second launch, no round trip
 * of the packed row: -5..-12 % against the two launches; q_packed and workspace are not touched and may be NULL).  From two rows on the
 * in-kernel quantiser measured behind the quantiser launch it replaces (profiles/r06_int4_front_probe.txt) and the call IS the two launches:
 * q_packed (uint8 [M, k_packed], caller-owned, else MIXQ_E_WORKSPACE) receives the packed rows, workspace as mixq_int4_fused_dequantize.
 * Same bits either way.  (The decode route that makes the 4-bit step 1.6 x the int8 one quantises in the fused RMSNorm in front:
 * mixq_rmsnorm_extract_quant4.) */
MIXQ_API int mixq_int4_linear_forward(const void* x, const uint8_t* B, void* x_scale, uint8_t* q_packed, const void* scale_col,
                                      const void* y, void* D, int M, int N, int k_packed, int epilogue, char* workspace, void* stream);
/* MI355X extension: A packed int4 [M, k_packed], B_int8 = the weight widened to int8 [N, 2 * k_packed] once at load time;
 * epilogue 0 = dequant, 1 = dequant + SiLU; workspace >= mixq_int4_fused_workspace_size(M, 0, k_packed) bytes (A only). */
MIXQ_API int mixq_int4_fused_dequantize_w8(const uint8_t* A, const int8_t* B_int8, const void* scale_row, const void* scale_col,
                                           const void* y, void* D, int M, int N, int k_packed, int epilogue, char* workspace,
                                           void* stream);
/* layernorm_forward_cuda_extract_outliers_int4 (layernorm/layernorm.cu:201-290, 379-414): the fused RMSNorm producer
 * with 4-bit rows: out fp16 [M,K] (normalised, outlier columns zeroed), outliers fp16 [M,len], q_packed uint8 [M,K/2]
 * (scale = fp16(amax/7)), scale fp16 [M].  Same constraints as mixq_rmsnorm_extract_quant. */
MIXQ_API int mixq_rmsnorm_extract_quant4(int M, int K, const void* x_f16, const void* gamma_f16, void* out_f16, float eps,
                                         const int32_t* ind, int len, void* outliers_f16, uint8_t* q_packed,
                                         void* scale_f16, void* stream);
/* unpack_int4_to_fp16 (cult.cu:3020-3043, 3088-3118): out[r, c] = fp16(int4 value of weight[r, ind[c]]),
 * weight uint8 [rows, cols_packed], out fp16 [rows, n]. */
MIXQ_API int mixq_unpack_int4_to_fp16(const uint8_t* weight, const int32_t* ind, int rows, int cols_packed, int n,
                                      void* out_f16, void* stream);
/* Sign-extending unpack of a packed int4 buffer (packed_bytes % 16 == 0) to int8 -- e.g. once per layer at load time. */
MIXQ_API int mixq_unpack_int4_to_int8(const uint8_t* src, int8_t* dst, size_t packed_bytes, void* stream);

/* ---- multi-GPU: rows of W sharded over the GPUs of a node --------------------------------------------- */
/* The path's ONE collective -- all-gather of the fp16 output columns, only where TP > 1 -- as one-sided peer writes over
 * xGMI (csrc/tp_kernels.hip).  The reference has no working counterpart (plugin.py:155-156 calls allreduce after an
 * N-split and is guarded off by `assert tp_size == 1`, tensorrt_llm/quantization/quantize.py:342).
 * One process per GPU.  Every rank allocates its destination buffers and its flag block with mixq_tp_buffer_alloc
 * (zeroed; returns the device pointer and a 64-byte hipIpcMemHandle_t), ships the handles to its peers over any host
 * channel (torch.distributed in parallel.PeerGather), and opens the peers' handles with mixq_tp_buffer_open: from then
 * on the peers' buffers are plain device pointers in this process.
 * mem_kind: memory that a REMOTE GPU writes while the local GPU reads / polls it must be FINEGRAINED or UNCACHED
 * (hipExtMallocWithFlags; what RCCL allocates its own flags and buffers with) -- system-scope atomics are only specified
 * on such allocations; COARSE (plain hipMalloc) is for single-GPU experiments only. */
#define MIXQ_TP_MEM_COARSE 0
#define MIXQ_TP_MEM_FINEGRAINED 1
#define MIXQ_TP_MEM_UNCACHED 2
#define MIXQ_TP_FLAG_WORDS 64 /* flag words per (buffer parity, producer) */
MIXQ_API int mixq_tp_buffer_alloc(size_t bytes, int mem_kind, void** dev_ptr, void* ipc_handle_64);
MIXQ_API int mixq_tp_buffer_open(const void* ipc_handle_64, void** dev_ptr);
MIXQ_API int mixq_tp_buffer_close(void* dev_ptr); /* a pointer from mixq_tp_buffer_open */
MIXQ_API int mixq_tp_buffer_free(void* dev_ptr);  /* a pointer from mixq_tp_buffer_alloc */
/* 64 bytes of host-mapped, coherent status words: [0] != 0 = a wait gave up (sticky), [1] = the sequence number it was
 * waiting for.  The host reads *host_ptr without synchronising the device; kernels get dev_ptr. */
MIXQ_API int mixq_tp_status_alloc(void** host_ptr, void** dev_ptr);
MIXQ_API int mixq_tp_status_free(void* host_ptr);
/* src fp16 [M, n_local] (this rank's operator output) -> columns [col0, col0 + n_local) of the fp16 [M, N] buffer of each
 * of the ndst <= 8 destinations (own rank included), then dst_flags[r][0 .. nflags) = seq (system scope, after a system
 * fence) for every r -- also when M == 0 (every consumer's wait of this call expects the flags).
 * done_counter: one zeroed device word of this rank (left zero).  n_local, N, col0 multiples of 8. */
MIXQ_API int mixq_tp_push_columns(const void* src, void* const* dst_bases, void* const* dst_flags, int ndst, int M,
                                  int n_local, int N, int col0, uint32_t seq, int nflags, void* done_counter,
                                  void* stream);
/* Makes `stream` wait until flags[r * MIXQ_TP_FLAG_WORDS + w] == seq for every producer r < nprod and word0 <= w < word0 +
 * nwords (this rank's flag block of the call's parity).  Gives up after patience_ms (0: 2000) and raises status[0]
 * (sticky: later waits return at once); trap_on_timeout != 0 additionally traps, so that nothing queued behind the wait
 * runs on a stale tensor (the stream's next synchronisation fails). */
MIXQ_API int mixq_tp_wait(const void* flags, int nprod, int word0, int nwords, uint32_t seq, void* status_dev,
                          int trap_on_timeout, uint32_t patience_ms, void* stream);

/* CAPTURABLE form of the stand-alone gather (round 5).  The reference's collectives are plugins INSIDE the engine, on the engine's
 * stream (tensorrt_llm/functional.py:3834-3880, called from plugin.py:155-156), so a TP step replays as one graph; the calls
 * above cannot be captured -- `seq` and the buffer parity are host state baked into the launch.  Here nothing of a call is host
 * state: `seq_word` (ONE zeroed device word per rank and gather object) holds the number of the last finished call, every launch
 * of call s reads s = *seq_word + 1 on the device, and mixq_tp_wait_seq stores s back.  Every rank owns ONE [M, N] destination
 * buffer (no parity), whose reuse is acknowledged explicitly.  A gather = three launches on `stream`, in this order:
 *   mixq_tp_arrive            this rank has reached call s (whatever read call s - 1's tensor is earlier in the stream): writes s
 *                             into peer_ack_words[r] (rank r's acknowledge word for THIS consumer: MIXQ_TP_FLAG_WORDS-strided block
 *                             of r, one word per consumer) for every r < npeer, then waits until own_ack_words[c * MIXQ_TP_FLAG_WORDS]
 *                             == s for every consumer c < npeer of OUR push
 *   mixq_tp_push_columns_seq  as mixq_tp_push_columns with seq = s, one flag word; status_dev (may be NULL) = the sticky status word of the
 *                             arrive / wait calls: once it is raised (an arrive timed out: some peer has not acknowledged that its single
 *                             destination buffer may be overwritten) the push stores NOTHING and publishes no flag, so that a slow peer
 *                             never reads a torn tensor -- its own wait of this call times out instead
 *   mixq_tp_wait_seq          waits for flags[r * MIXQ_TP_FLAG_WORDS] == s for every producer r < nprod, then *seq_word = s
 * The three may be captured in a HIP graph, replayed any number of times and mixed with eager calls on the same objects.
 * Time-outs as mixq_tp_wait (sticky status word; optional trap). */
MIXQ_API int mixq_tp_arrive(void* const* peer_ack_words, const void* own_ack_words, int npeer, const void* seq_word,
                            void* status_dev, int trap_on_timeout, uint32_t patience_ms, void* stream);
MIXQ_API int mixq_tp_push_columns_seq(const void* src, void* const* dst_bases, void* const* dst_flags, int ndst, int M,
                                      int n_local, int N, int col0, const void* seq_word, void* done_counter, const void* status_dev,
                                      void* stream);
MIXQ_API int mixq_tp_wait_seq(const void* flags, int nprod, void* seq_word, void* status_dev, int trap_on_timeout,
                              uint32_t patience_ms, void* stream);

/* The same all-gather FUSED INTO THE OPERATOR (round 3): mixq_enqueue's prefill path with the GEMM's store path writing
 * every finished block straight into this rank's column block [col0, col0 + N_local) of all ndst destination buffers
 * ([M, n_total] fp16, own rank included) -- no [M, N_local] output, no push launch, no second read of the block; the
 * transfer runs under the GEMM.  Flag words: the M rows are cut into mixq_tp_flag_words(M) chunks (<= 64; whole groups of
 * 256-row tile rows in the kernel's walk order) and word c of dst_flags[r] receives `seq` when the last tile of chunk c
 * has been acknowledged by every destination, so a consumer may start on chunk c (mixq_tp_wait word0 = c) while later
 * chunks are still being computed; waiting for words [0, mixq_tp_flag_words(M)) waits for the whole tensor.
 * counters: MIXQ_TP_FLAG_WORDS zeroed device words of this rank (left zero).  inputs / inputDesc / workspace as
 * mixq_enqueue (inputs[5], [6] unused).  Only shapes that take the plain 256 x 256 ping-pong kernel are served
 * (mixq_tp_fused_supported); others return MIXQ_E_SHAPE before anything is launched: use mixq_enqueue +
 * mixq_tp_push_columns.  Cannot be timed on one GPU (the 2-process IPC test covers the protocol, not xGMI). */
typedef struct mixq_tp_epilogue {
    int32_t ndst;          /* destinations, 1..8 */
    int32_t n_total;       /* row stride of every destination in elements */
    int32_t col0;          /* this rank's first column */
    uint32_t seq;          /* sequence number of the call */
    void* dst_bases[8];    /* [M, n_total] fp16 buffers (fine-grained / uncached: mixq_tp_buffer_alloc) */
    void* dst_flags[8];    /* this producer's MIXQ_TP_FLAG_WORDS flag words in each destination */
    void* counters;        /* MIXQ_TP_FLAG_WORDS zeroed uint32 of this rank */
} mixq_tp_epilogue;
MIXQ_API int mixq_tp_fused_supported(int64_t M, int64_t N_local, int64_t K);
MIXQ_API int mixq_tp_flag_words(int64_t M);
MIXQ_API int mixq_enqueue_tp(const mixq_handle* h, const mixq_tensor_desc* inputDesc, const void* const* inputs,
                             void* workspace, const mixq_tp_epilogue* tp, void* stream);

/* ---- host helpers ----------------------------------------------------------------------------- */
/* preprocess_weights (weightonlykernel/cutlass_kernels/cutlass_preprocessors.cc:536-545), int8, arch 80-90:
 * row-major int8 [rows=K, cols=N] -> interleaved uint8.  Host memory.  And its inverse. */
MIXQ_API int mixq_preprocess_weights_int8(uint8_t* preprocessed, const int8_t* row_major, size_t rows, size_t cols);
MIXQ_API int mixq_unprocess_weights_int8(int8_t* row_major, const uint8_t* preprocessed, size_t rows, size_t cols);

/* ---- plan introspection (MI355X extension; host only, needs no GPU) ---------------------------------------------------------
 * What mixq_enqueue would launch for a call with M rows on an [N, K] layer (128 outlier columns, workspace sized by
 * mixq_workspace_size, knobs at their defaults), written as text into buf: the kernel family, its tile / split / weight route.  The
 * selection is a set of tables fitted on measurements; this entry makes them readable from the outside -- an integrator
 * can list the forms its model's shapes will take, and tests/golden/selection_table.json pins the answers on BASELINE.json's shapes so
 * that a rule change shows up as a diff of that table.  have_weight_image != 0: as if `weight` had a registered image.
 * Returns MIXQ_OK, or MIXQ_E_BADARG.  Without a device the 256-CU tables answer. */
MIXQ_API int mixq_describe_plan(int64_t M, int64_t N, int64_t K, int have_weight_image, char* buf, size_t len);

/* ---- debug / measurement (NOT FOR PRODUCTION) ----------------------------------------------------------------------------
 * Everything named mixq_debug_* is PROCESS-GLOBAL state (atomics: race-free, but a caller that flips a knob changes the
 * kernel selection of every thread and stream of the process) and exists for the tests, the A/B tools under tools/ and the
 * in-kernel timelines.  The operator boundary above holds no such state: with the knobs at their defaults --
 * mixq_debug_reset() -- the selection is a pure function of (M, N, K, scratch).  No knob changes results beyond what the
 * tests pin (every form is bit-identical or within the stated tolerance), except the ablation ranges marked "wrong results". */
/* The set_* knobs below act ONLY in a process whose environment holds MIXQ_DEBUG_KNOBS=1 when the first of them is called (read
 * once); anywhere else they are ignored, so that no code path of a production process can change the kernel selection.
 * mixq_debug_knobs_enabled() says which kind of process this is.  mixq_debug_reset() and the reporting entries always work. */
MIXQ_API int mixq_debug_knobs_enabled(void);
MIXQ_API void mixq_debug_reset(void);
/* Test / measurement knob: main-loop schedule of the fused GEMM.  0 = auto (default), 1 = 2-barrier double-buffered
 * kernel only, 2 = 256x256 ping-pong kernel for every M > 4.  Results are identical bit for bit in all modes. */
MIXQ_API void mixq_debug_set_gemm_variant(int variant);
/* Measurement knob: when non-NULL, every workgroup of the ping-pong GEMM launched by mixq_gemm_mixed writes 8 uint64
 * shader-clock stamps (start, prologue done, main loop done, outlier operands staged, dequant math done, tile staged,
 * stores issued, stores drained) to buffer[block*8 ..].  NULL (default) disables it. */
MIXQ_API void mixq_debug_set_stamp_buffer(void* device_u64_8_per_block);
/* The same for the quantiser launch of mixq_enqueue / mixq_quant_extract and (through the buffer above) the skinny GEMM:
 * 100 MHz wall-clock stamps (one time base across kernels): quantiser = entry, row loads issued, gather issued, amax
 * reduced, stores issued, stores acknowledged; skinny GEMM = entry, epilogue operands requested, first 16 k-steps done,
 * last MFMA, LDS hand-over, stores issued, stores acknowledged (tools/small_m_timeline.py). */
MIXQ_API void mixq_debug_set_quant_stamp_buffer(void* device_u64_8_per_block);
/* Reporting only: name of the kernel family the fused GEMM selected on its most recent launch in this process. */
MIXQ_API const char* mixq_debug_last_gemm_kernel(void);
MIXQ_API const char* mixq_version(void);
/* ABI revision of THIS header.  It is bumped whenever an exported entry changes its argument list in place (revision 3: the
 * mixq_tp_buffer_alloc / mixq_tp_push_columns / mixq_tp_wait signatures of round 3), so that a host built against an older
 * header can refuse to run instead of passing shifted arguments: `if (mixq_abi_version() != MIXQ_ABI_VERSION) fail`.  The
 * reference-named entries (initOpenAiTritonPlugins, the plugin lifecycle, mixq_enqueue) have not changed since revision 1. */
#define MIXQ_ABI_VERSION 4
MIXQ_API int mixq_abi_version(void);
MIXQ_API const char* mixq_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif /* MIXQ_H_ */
