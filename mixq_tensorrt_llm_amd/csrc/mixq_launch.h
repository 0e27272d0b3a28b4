// Host-visible launch interface between mixq_api.hip (C ABI, orchestration) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace mixq {

// Per-device once-flags: attributes set with hipFuncSetAttribute (and device symbol addresses, CU counts) belong to the
// CURRENT device, and a process may drive several GPUs.  One bit per device ordinal (ordinals >= 64 are simply redone).
inline int current_device()
{
    int d = 0;
    return hipGetDevice(&d) == hipSuccess && d >= 0 ? d : 0;
}
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    bool done(int dev) const { return dev < 64 && ((mask.load(std::memory_order_acquire) >> dev) & 1u); }
    void set(int dev)
    {
        if (dev < 64) mask.fetch_or(uint64_t{1} << dev, std::memory_order_release);
    }
};
// hipFuncAttributeMaxDynamicSharedMemorySize for `kern` on the current device, once per device (idempotent if raced).
template <class Kern>
inline hipError_t ensure_dynamic_lds(Kern kern, size_t bytes, DeviceOnce& once)
{
    const int dev = current_device();
    if (once.done(dev)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)bytes);
    if (e == hipSuccess) once.set(dev);
    return e;
}
int num_cus(); // of the current device (cached per device)

enum { EPI_DEQUANT = 0, EPI_DEQUANT_SILU = 1, EPI_INT32 = 2, EPI_DEQUANT_SILU_MUL = 3,
       EPI_F16GEMM = 4 }; // 4: the operands are fp16 (K counts BYTES), fp32 accumulation, D = fp16(A B^T): the ping-pong kernel
                          // with v_mfma_f32_32x32x16_f16 (second pass of the two-pass fpA_intB GEMM, w8a16_gemm_kernels.hip)
constexpr bool epi_has_silu(int epi) { return epi == EPI_DEQUANT_SILU || epi == EPI_DEQUANT_SILU_MUL; }

// TP all-gather fused into the GEMM's store path (gemm_pp_kernels.hip, TP = true): every finished 32-row block of a wave
// tile is written straight into this rank's column block of EVERY destination's [M, n_total] buffer (one-sided peer writes
// over xGMI, write-through at system scope), and when the last tile of an M chunk (`chunk_tile_rows` rows of 256 x 256
// tiles) has been acknowledged the chunk's flag word in every destination receives `seq`.  No separate push launch, no
// second read of the [M, N/tp] block, and the transfer runs under the GEMM itself.
constexpr int kTpMaxPeers = 8;
constexpr int kTpFlagWords = 64; // flag words per (parity, producer): word 0 = stand-alone push, one per M chunk when fused
struct TpEpilogue {
    int ndst;                      // 0: off
    int ldd;                       // row stride of every destination in elements (= n_total)
    void* base[kTpMaxPeers];       // destination r's [M, n_total] fp16 buffer + this rank's column offset
    unsigned* flag[kTpMaxPeers];   // this producer's kTpFlagWords flag words in destination r
    unsigned seq;
    unsigned* counters;            // this rank's per-chunk tile counters (kTpFlagWords words, zero; left zero)
    int chunk_tile_rows;           // tile rows (of 256 rows) per flag word
};
int tp_chunk_tile_rows(int M);     // tile rows per flag word for an M-row call: a multiple of the kernel's GROUP_M walk
int tp_flag_words(int M);          // flag words a consumer must wait for = ceil(tile rows / chunk_tile_rows) <= kTpFlagWords
bool gemm_tp_fused_supported(int M, int N, int K, int O); // launch_gemm would take the plain 256 x 256 ping-pong kernel

struct GemmParams {
    const int8_t* A;      // qA [M,K]
    const int8_t* B;      // W  [N,K]
    const uint16_t* sA;   // [M]
    const uint16_t* sW;   // [N]
    const uint16_t* fpA;  // [M,O] or null
    const uint16_t* fpW;  // [N,O] or null
    const uint16_t* Y;    // fp16 [M,N] addend or null (int8FusedDequantize API)
    const uint16_t* Mul;  // fp16 [M,N] multiplicand of EPI_DEQUANT_SILU_MUL: D = fp16(fp16(silu(..)) * Mul), else null
    void* D;              // fp16 [M,N] (EPI_DEQUANT*) or int32 [M,N] (EPI_INT32)
    const void* zeros;    // >= 16 B of device zeros (K / O tails)
    int M, N, K, O;
    int a_frag;           // layout of A: 0 row-major | 1 the skinny GEMM's fragment-major image (only valid when gemm_takes_skinny) |
                          // 2 K-slice-major [K/128][M][128 B] (read by the plain 256 x 256 ping-pong kernel only; no producer ships:
                          // measured a net loss, docs/LAB_NOTEBOOK.md R3.10); 1 is written by mixq_enqueue's quantiser (FRAG)
    const void* b_image;  // the registered fragment-major image of B this call may stream (resolved once per call by the API layer), or null
    int b_frag;           // (set by the skinny launcher from b_image, a_frag == 1 only) layout of B: 0 row-major [N,K] | 1 fragment-major image
                          // (1-KiB blocks [16-feature tile][64-byte k-step], lane l = feature % 16 + 16 * (k / 16 % 4) at l * 16) | 2 the same, non-temporal loads
    void* dbg;            // measurement only: 8 x uint64 s_memtime stamps per block (ping-pong kernel), else null
    int flags;            // (set by launch_gemm_deep) bit 0: measurement -- every wave issues its slice copies before its MFMAs
    int xsplit;           // (set by launch_epi) workgroups per tile of the small-tile kernels' K split, else 0
    int splitk_solo;      // (set by launch_gemm_pp_splitk) leading tiles that are not split
    unsigned splitk_patience; // (set by launch_gemm_pp_splitk) wall-clock ticks a workgroup waits for its partners before
                              // it defers its share to the last arriver
    TpEpilogue tp;        // (launch_gemm_pp_tp only) peer-write epilogue, else ndst == 0
    void* splitk_ws;      // device scratch of gemm_splitk_workspace_size bytes whose arrival words are zero, or null: lets
                          // launch_gemm split K over 2 / 4 workgroups per 256x256 tile when the tiles alone cover at
                          // most half / a quarter of the CUs (gemm_pp_kernels.hip)
};

hipError_t launch_gemm(const GemmParams& p, int epi, hipStream_t st);
void describe_gemm_plan(const GemmParams& p, int epi, char* buf, size_t len); // what launch_gemm would launch, as text (host only)
int gemm_deep_factor(int M, int N, int K, bool have_scratch);                 // workgroups per 128 x 128 tile of the deep form, 0: not taken
int skinny_weight_route(int a_frag, bool image, int M, int N, int K);         // GemmParams::b_frag the skinny launcher would set
hipError_t launch_gemm_pp(const GemmParams& p, int epi, hipStream_t st);  // 256x256 ping-pong schedule
hipError_t launch_gemm_pp128(const GemmParams& p, int epi, hipStream_t st);
hipError_t launch_gemm_pp_tp(const GemmParams& p, hipStream_t st); // EPI_DEQUANT with p.tp.ndst destinations instead of p.D
// D[M,N] fp16 = A[M,K] B[N,K]^T, fp16 operands (K % 8 == 0, N % 8 == 0, 16-byte aligned rows), fp32 accumulation
hipError_t launch_gemm_f16_pp(const void* A, const void* B, void* D, int M, int N, int K, const void* zeros, hipStream_t st);
hipError_t launch_gemm_f16_pp128(const void* A, const void* B, void* D, int M, int N, int K, const void* zeros, hipStream_t st); // 128 x 256 tiles // 128x256 ping-pong schedule (mid-size problems)
bool gemm_pp128_wins(int M, int N, int K); // launch_gemm's rule for taking the 128x256 tiles (then no K split, no scratch)
constexpr size_t kSplitkWordsBytes = 16384; // hand-over words (64 B per tile) of up to 256 split tiles, at the start of the scratch
struct SplitPlan {
    int s;    // workgroups per split tile: 0 (no split form for this shape) / 2 / 4 / 8
    int solo; // leading tiles (whole waves) computed by one workgroup each, inside the same launch
};
SplitPlan gemm_splitk_plan(int M, int N, int K);
int gemm_splitk_factor(int M, int N, int K);              // = plan.s
size_t gemm_splitk_workspace_size(int M, int N, int K);   // 0: the shape does not use the split form (or it is off)
size_t gemm_splitk_workspace_bound();                     // max of the above over all shapes (for workspace sizing)
hipError_t launch_gemm_pp_splitk(const GemmParams& p, int epi, hipStream_t st);
void set_splitk_patience(unsigned ticks); // test knob: 0 = every workgroup but the last arriver defers at once
void set_splitk_force(int v); // -1 automatic (default), 0 off, 2 / 4: that factor wherever the shape allows it
// the same idea for the small-tile kernels with in-workgroup split (gemm_kernels.hip, XS): few tiles, long K
int gemm_xsplit_factor(int M, int N, int K);              // 0 / 2 / 4 / 8 / 16
size_t gemm_xsplit_workspace_size(int M, int N, int K);
size_t gemm_xsplit_workspace_bound();
void set_xsplit_force(int v); // -1 automatic (default), 0 off, 2 / 4 / 8 / 16 forced
// mid-M deep form (gemm_kernels.hip): 128 x 128 tiles, 4 stages in flight, K split over 1 / 2 / 4 / 8 workgroups per tile
bool gemm_deep_takes(int M, int N, int K, bool have_scratch);
size_t gemm_deep_workspace_size(int M, int N, int K); // 0: the form is not used for this shape, or needs no scratch
void set_deep_force(int v);
hipError_t launch_gemm_deep(const GemmParams& p, int epi, hipStream_t st);
// round-6 schedule of the same tiles (gemm_mid_kernels.hip): 2 copy-only waves + 8 compute waves, one barrier per pair of slices; p.xsplit workgroups per tile
hipError_t launch_gemm_mid(const GemmParams& p, int epi, hipStream_t st);
void set_mid_rot(int mode);  // measurement knob 1413 (default: 3 = by rule) / 1411 (0 = never) / 1410 (1 = unsplit tiles) / 1412 (2 = always)
void set_mid_bn(int mode);   // measurement knob 1430 (default: 0 = by rule) / 1431 (1 = 128-wide tiles always)
int gemm_mid_tile_width(int M, int N, int xsplit); // 128 | 96: the tile width launch_gemm_mid takes
bool gemm_skinny_supported(const GemmParams& p);
hipError_t launch_gemm_skinny(const GemmParams& p, int epi, hipStream_t st); // M <= 64: GEMV-like, HBM-bound on W
hipError_t launch_gemm_pp_ablate(const GemmParams& p, int abl, hipStream_t st);  // timing experiments only
void set_gemm_variant(int v);
bool w8a16_skinny_takes(int M, int N, int K); // the fpA_intB skinny form serves this shape (w8a16_gemm_kernels.hip)
// fpA_intB measurement / test knobs (mixq_debug_set_gemm_variant 80..89, 831..834, 840..844, 850..858, 800 + ABL); `form`:
//   -1 everything automatic | 0 narrow form | 1..4 wide form with 32 / 64 / 128 / 256-row tiles | 100 + ABL ablation of the
//   256-row form | 200 / 201 / 202 two-pass form automatic / never / always, 203 / 204 its second pass on 256- / 128-row tiles
//   | 300 / 301 skinny form automatic / off, 302..305 one of its shapes | 306 / 307 / 308 1..4 tokens through the skinny form
//   always / never / automatic;  `ks`: -2 keep, -1 automatic, n = workgroups per tile along K.
// fpA_intB measurement / test knobs (mixq_debug_set_gemm_variant 80..89, 831..834, 840..844, 850..858, 800 + ABL); `form`:
//   -1 everything automatic | 0 narrow form | 1..4 wide form with 32 / 64 / 128 / 256-row tiles | 100 + ABL ablation of the
//   256-row form | 200 / 201 / 202 two-pass form automatic / never / always, 203 / 204 its second pass on 256- / 128-row tiles
//   | 300 / 301 skinny form automatic / off, 302..305 one of its shapes | 306 / 307 / 308 1..4 tokens through the skinny form
//   always / never / automatic;  `ks`: -2 keep, -1 automatic, n = workgroups per tile along K.
void set_wo_force(int form, int ks);
const char* last_gemm_kernel(); // kernel family launch_gemm chose last (reporting only)
void set_skinny_kw(int kw); // measurement knob: K-split width of the skinny kernel (0 = auto)
void set_skinny_nt(int nt);
void set_skinny_wrows(int on); // knob 884 on (default) / 885 off: row-major weights read in 256-byte runs (gemm_skinny_kernels.hip WFRAG == 3)
void set_skinny_wfrag(int mode); // knob 880 automatic | 881 weight images with plain loads | 882 with non-temporal loads | 883 images ignored
// weight images (gemm_skinny_kernels.hip): a fragment-major copy of an int8 [N, K] weight, registered under the weight's pointer
hipError_t launch_weight_image(const int8_t* W, int8_t* img, int N, int K, hipStream_t st);
hipError_t register_weight_image(const void* weight, const void* image, int N, int K, hipStream_t st); // records the weight's content tag; synchronises st
bool unregister_weight_image(const void* weight);
const void* resolve_weight_image(const void* weight, int N, int K, hipStream_t st); // once per call; verifies the content tag on first use
int verify_weight_image(const void* weight, hipStream_t st);  // 1 current | 0 stale (dropped) | -1 nothing registered; synchronises st
int weight_image_stale_count();
int skinny_feature_tiles(int M, int N, int K, bool image = false); // (image: the call streams a registered weight image) measurement knob 894 / 895 / 896: feature tiles per workgroup of the fragment-major form auto / 1 / 2
hipError_t launch_gemm_fp16(const void* fpA, const void* fpW, void* Out, int M, int N, int O, hipStream_t st);
hipError_t launch_dequantization(void* out, const int32_t* x, const void* sRow, const void* sCol, int M, int N,
                                 hipStream_t st);
hipError_t launch_dequantization_silu(void* out, const int32_t* x, const void* sRow, const void* sCol, const void* y,
                                      int M, int N, hipStream_t st);
hipError_t launch_quant_extract(void* A, int8_t* qA, void* sA, void* fpA, const int32_t* ind, int M, int K, int O,
                                bool zero, hipStream_t st,
                                void* zero_words = nullptr, // (kSplitkWordsBytes to clear on the way, or null)
                                int frag = 0); // qA layout: 0 row-major | 1 the skinny GEMM's fragment order (quant_frag_layout_supported)
void set_quant_block_rows(int m); // measurement knob 1300 (rules) / 1301..1312 (64 << n rows)
int quant_block_rows(int nvec, bool norm_producer); // rows up to which a whole 256-thread block takes one row of nvec 16-byte vectors
bool quant_frag_layout_supported(int M, int K);   // the quantiser can write the fragment-major image for this shape
bool gemm_takes_skinny(const GemmParams& p, int epi);
bool qa_frag_enabled(); // test knob 890 / 891 (default on)
bool probe_a2();        // timing probe 898 / 899 // launch_gemm would run gemm_skinny_kernel on this problem
void set_quant_stamp_buffer(void* device_u64_8_per_block); // measurement only (NULL in production)
hipError_t launch_quant_with_scale(const void* src, const void* scale, int8_t* dst, int M, int K, hipStream_t st);
hipError_t launch_extract(void* A, void* fpA, const int32_t* ind, int M, int K, int O, bool zero, hipStream_t st);
hipError_t launch_rmsnorm_quant(const void* X, const void* gamma, void* out, void* outl, const int32_t* ind, int8_t* q,
                                void* scale, float eps, int M, int K, int O, int quant /* 0, 8, 4 */, hipStream_t st,
                                int q_layout = 0 /* 1: fragment-major int8 rows (quant == 8, quant_frag_layout_supported) */);
hipError_t launch_find_outliers(const void* A, int M, int K, float sigma, unsigned* mask, int32_t* ind, int32_t* count,
                                int capacity, hipStream_t st);
hipError_t launch_dequant_columns(const int8_t* W, const void* sW, const int32_t* ind, int len, void* out, int N, int K,
                                  hipStream_t st);
hipError_t launch_quant4_rows(const void* A, uint8_t* q, void* sA, int M, int K, hipStream_t st);
hipError_t launch_unpack_s4(const uint8_t* src, int8_t* dst, size_t packed_bytes, hipStream_t st);
// packed-int4 weight stream (int4_gemm_kernels.hip): p.A / p.B packed int4 [M, K] / [N, K], p.K = PACKED bytes per row, p.Y addend or null
void set_s4_wrows(int mode); // knob 872 by rule (default) / 873 off / 874 always: packed weights read in 256-byte runs (int4_gemm_kernels.hip WROWS)
bool gemm_skinny_s4_supported(int M, int N, int k_packed);
hipError_t launch_gemm_skinny_s4(const GemmParams& p, int epi, hipStream_t st);
// the same with the fp16 rows quantised inside the launch (M <= 16): p.A = fp16 [M, 2 p.K], p.sA = the row scales, WRITTEN (workgroup 0)
bool gemm_skinny_s4q_supported(int M, int N, int k_packed);
hipError_t launch_gemm_skinny_s4q(const GemmParams& p, int epi, hipStream_t st);
hipError_t launch_unpack_s4_columns(const uint8_t* weight, const int32_t* ind, int rows, int cols_packed, int n,
                                    void* out, hipStream_t st);
hipError_t launch_w8a16(const void* A, const uint8_t* Wq, const void* scale, void* Out, int M, int N, int K,
                        hipStream_t st);

void note_gemm_kernel(const char* name);
// TP all-gather of the output columns as one-sided peer writes + flags (tp_kernels.hip)
// seq_word != null (capturable form): the call number is *seq_word + 1, read on the device; launch_tp_wait stores it back
hipError_t launch_tp_push(const void* src, void* const* dst_bases, unsigned* const* dst_flags, int ndst, int M, int n_loc,
                          int N, int col0, unsigned seq, int nflags, unsigned* done_counter, hipStream_t st,
                          const unsigned* seq_word = nullptr,
                          const unsigned* status = nullptr); // (capturable form) the sticky status word: raised -> nothing is stored or published
hipError_t launch_tp_wait(const unsigned* flags, int nprod, int word0, int nwords, unsigned seq, unsigned* status, int trap,
                          unsigned patience_ms, hipStream_t st, unsigned* seq_word = nullptr);
hipError_t launch_tp_arrive(unsigned* const* peer_acks, const unsigned* own_acks, int npeer, const unsigned* seq_word,
                            unsigned* status, int trap, unsigned patience_ms, hipStream_t st);
// fpA_intB GEMM (M > 4) on the interleaved qweight (w8a16_gemm_kernels.hip); scratch may be null (no K split)
size_t w8a16_gemm_workspace_size(int M, int N, int K);
hipError_t launch_w8a16_gemm(const void* A, const uint8_t* Wq, const void* scale, void* Out, int M, int N, int K,
                             void* scratch, size_t scratch_bytes, const void* zeros, hipStream_t st);

} // namespace mixq
