// Host side of libmixq_mi355x.so: plugin object, creator registry, enqueue orchestration and the C ABI declared in
// include/mixq.h.  Mirrors the host half of the reference (TsinghuaMixQPlugin.{h,cpp}, MixQPlugins.cpp) without
// TensorRT: there is no nvinfer on MI355X, so the "plugin" is a plain C++ object behind an opaque C handle.
#include "../../include/mixq.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_set>
#include <vector>

#include "mixq_launch.h"

namespace mixq {
// 256 bytes of device zeros: source of the K / O tail chunks of the staging loads (never written).
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

static const void* zero_page() // address of the symbol on the CURRENT device (a process may drive several GPUs)
{
    static std::atomic<const void*> cache[64];
    const int dev = current_device();
    const void* ptr = dev < 64 ? cache[dev].load(std::memory_order_relaxed) : nullptr;
    if (!ptr) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_zero_page)) != hipSuccess) p = nullptr;
        ptr = p;
        if (dev < 64) cache[dev].store(ptr, std::memory_order_relaxed);
    }
    return ptr;
}
} // namespace mixq

namespace {

constexpr int kNumOutliers = 128;      // TsinghuaMixQPlugin.cpp:518, plugin.py:102-105
constexpr int kSmallMFastPath = 4;     // TsinghuaMixQPlugin.cpp:472, fpA_intB_gemm_wrapper.h:4
constexpr size_t kWorkspaceAlign = 128; // kCudaMemAlign, TsinghuaMixQPlugin.cpp:204

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int hip_rc(hipError_t e) { return e == hipSuccess ? MIXQ_OK : MIXQ_E_HIP; }

} // namespace

struct mixq_handle {
    int32_t mm, mn, mk; // serialized state: exactly the reference's three ints (TsinghuaMixQPlugin.cpp:813-820)
    std::string ns;
    bool initialized = false;
};

// ------------------------------------------------------------------------------------------- registry ---
namespace {
class CreatorRegistry { // TritonPluginCreatorRegistry, MixQPlugins.cpp:33-114
public:
    static CreatorRegistry& instance()
    {
        static CreatorRegistry r;
        return r;
    }
    bool add(const char* ns)
    {
        std::lock_guard<std::mutex> lock(mu_);
        keys_.insert(key("MixQ", "1", ns ? ns : ""));
        return true;
    }
    bool has(const char* name, const char* version, const char* ns)
    {
        std::lock_guard<std::mutex> lock(mu_);
        return keys_.count(key(name ? name : "", version ? version : "", ns ? ns : "")) != 0;
    }

private:
    static std::string key(const char* name, const char* version, const char* ns)
    {
        return std::string(ns) + "::" + name + " version " + version;
    }
    std::mutex mu_;
    std::unordered_set<std::string> keys_;
};
} // namespace

extern "C" {

bool initOpenAiTritonPlugins(void* /*logger*/, const char* libNamespace)
{
    try {
        return CreatorRegistry::instance().add(libNamespace);
    } catch (...) {
        return false;
    }
}

int mixq_registry_has_creator(const char* name, const char* version, const char* libNamespace)
{
    try {
        return CreatorRegistry::instance().has(name, version, libNamespace) ? 1 : 0;
    } catch (...) {
        return 0;
    }
}

const char* mixq_plugin_type(void) { return "MixQ"; }
const char* mixq_plugin_version(void) { return "1"; }
static std::atomic<void*> g_dbg_stamps{nullptr}; // measurement knob only (NULL in production)
// The measurement knobs are PROCESS-GLOBAL, so a production process must not be one stray call away from another kernel
// selection: they act only in a process that opted in with MIXQ_DEBUG_KNOBS=1 in its environment BEFORE the first
// knob call (read once).  mixq_debug_reset() and the reporting entries are always allowed.
static bool debug_knobs_enabled()
{
    static const bool on = [] {
        const char* e = getenv("MIXQ_DEBUG_KNOBS");
        return e != nullptr && e[0] == '1';
    }();
    return on;
}
int mixq_debug_knobs_enabled(void) { return debug_knobs_enabled() ? 1 : 0; }

void mixq_debug_set_stamp_buffer(void* device_u64_8_per_block)
{
    if (debug_knobs_enabled() || device_u64_8_per_block == nullptr) g_dbg_stamps.store(device_u64_8_per_block);
}

void mixq_debug_set_quant_stamp_buffer(void* device_u64_8_per_block)
{
    if (debug_knobs_enabled() || device_u64_8_per_block == nullptr) mixq::set_quant_stamp_buffer(device_u64_8_per_block);
}

static void set_int4_stream(int on);
static void set_int4_fuse_quant(int on);
void mixq_debug_set_gemm_variant(int variant)
{
    if (!debug_knobs_enabled()) return;
    if (variant >= 875 && variant <= 877) { // mixq_int4_linear_forward: rows quantised inside the stream launch by the measured rule (875, default: one row) / never (876) / whenever the kernel serves the size (877: tests)
        set_int4_fuse_quant(variant == 875 ? 1 : variant == 876 ? 0 : 2);
        return;
    }
    if (variant >= 872 && variant <= 874) { // packed-int4 weight stream, 256-byte runs: 872 by the measured rule (default), 873 off, 874 always
        mixq::set_s4_wrows(variant == 872 ? 1 : variant == 873 ? 0 : 2);
        return;
    }
    if (variant == 870 || variant == 871) { // packed-int4 weight stream for decode batches: 870 on (default), 871 off (unpack route)
        set_int4_stream(variant == 870);
        return;
    }
    mixq::set_gemm_variant(variant < 0 ? 0 : variant);
}

void mixq_debug_reset(void)
{
    // every knob of the mixq_debug_* family back to the production default (see include/mixq.h "debug / measurement")
    g_dbg_stamps.store(nullptr);
    mixq::set_quant_stamp_buffer(nullptr);
    for (int v : {-1 /* schedule (back to the MIXQ_GEMM_VARIANT default), tile configuration, skinny K width */, 79 /* K splits over workgroups automatic */, 69, 65, 91,
                  80 /* fpA_intB forms automatic */, 85, 840, 843, 848 /* non-temporal loads of large weights on */, 850, 858, 891 /* fragment-major qA on */, 893 /* skinny range: the rule */,
                  894 /* feature tiles automatic */, 880 /* skinny-GEMM weight image off */, 884 /* row-major weights in 256-byte runs */, 1240 /* mid-M deep form automatic */, 1238, 1413 /* its K walk rotated by the measured rule */, 1420 /* the deep plan takes it from 129 rows on */, 1430 /* its tile width by rule */, 1290 /* non-temporal weight copies of single-row tile launches: by rule */, 1300 /* quantisers: block-per-row by the measured rules */})
        mixq::set_gemm_variant(v);
    set_int4_stream(1);
    set_int4_fuse_quant(1);
    mixq::set_s4_wrows(1);
}

const char* mixq_debug_last_gemm_kernel(void) { return mixq::last_gemm_kernel(); }

const char* mixq_version(void) { return "mixq-mi355x 0.3 (gfx950)"; }
int mixq_abi_version(void) { return MIXQ_ABI_VERSION; }

const char* mixq_error_string(int code)
{
    switch (code) {
    case MIXQ_OK: return "ok";
    case MIXQ_E_BADARG: return "bad argument";
    case MIXQ_E_SHAPE: return "unsupported shape";
    case MIXQ_E_ALIGN: return "pointer not 16-byte aligned";
    case MIXQ_E_HIP: return "HIP error";
    case MIXQ_E_WORKSPACE: return "workspace missing";
    case MIXQ_E_STALE: return "registered weight image does not match the weight's current content";
    default: return "unknown";
    }
}

// ------------------------------------------------------------------------------------------ lifecycle ---
mixq_handle* mixq_create(int32_t m, int32_t n, int32_t k)
{
    mixq_handle* h = new (std::nothrow) mixq_handle;
    if (!h) return nullptr;
    h->mm = m, h->mn = n, h->mk = k;
    return h;
}

mixq_handle* mixq_create_from_fields(const mixq_plugin_field* fields, int32_t nbFields)
{
    // createPlugin parses "m","n","k" (TsinghuaMixQPlugin.cpp:906-919) although the creator advertises
    // "mm","mn","mk" (:873-875); unknown names are ignored and missing ones stay 0, as in the reference.
    int32_t m = 0, n = 0, k = 0;
    if (nbFields > 0 && !fields) return nullptr;
    for (int i = 0; i < nbFields; ++i) {
        const mixq_plugin_field& f = fields[i];
        if (!f.name || !f.data) continue;
        int32_t v;
        std::memcpy(&v, f.data, sizeof(v));
        if (!std::strcmp(f.name, "m")) m = v;
        else if (!std::strcmp(f.name, "n")) n = v;
        else if (!std::strcmp(f.name, "k")) k = v;
    }
    return mixq_create(m, n, k);
}

const mixq_plugin_field* mixq_get_field_names(int32_t* nbFields)
{
    // TsinghuaMixQPlugin.cpp:868-878: PluginField("mm" / "mn" / "mk", nullptr, PluginFieldType::kINT32, -1)
    static const mixq_plugin_field kFields[3] = {{"mm", nullptr, MIXQ_FIELD_INT32, -1},
                                                 {"mn", nullptr, MIXQ_FIELD_INT32, -1},
                                                 {"mk", nullptr, MIXQ_FIELD_INT32, -1}};
    if (nbFields) *nbFields = 3;
    return kFields;
}

mixq_handle* mixq_deserialize(const void* data, size_t length)
{
    if (!data || length < 3 * sizeof(int32_t)) return nullptr;
    int32_t v[3];
    std::memcpy(v, data, sizeof(v));
    return mixq_create(v[0], v[1], v[2]);
}

size_t mixq_serialization_size(const mixq_handle*) { return 3 * sizeof(int32_t); }

void mixq_serialize(const mixq_handle* h, void* buffer)
{
    if (!h || !buffer) return;
    const int32_t v[3] = {h->mm, h->mn, h->mk};
    std::memcpy(buffer, v, sizeof(v));
}

mixq_handle* mixq_clone(const mixq_handle* h)
{
    if (!h) return nullptr;
    mixq_handle* c = new (std::nothrow) mixq_handle(*h);
    return c;
}

void mixq_destroy(mixq_handle* h) { delete h; }

int mixq_initialize(mixq_handle* h)
{
    if (!h) return MIXQ_E_BADARG;
    h->initialized = true; // the reference creates a cuBLAS handle here; nothing to create on this path
    return MIXQ_OK;
}

void mixq_terminate(mixq_handle* h)
{
    if (h) h->initialized = false;
}

int mixq_get_mnk(const mixq_handle* h, int32_t* m, int32_t* n, int32_t* k)
{
    if (!h) return MIXQ_E_BADARG;
    if (m) *m = h->mm;
    if (n) *n = h->mn;
    if (k) *k = h->mk;
    return MIXQ_OK;
}

int mixq_set_namespace(mixq_handle* h, const char* ns)
{
    if (!h) return MIXQ_E_BADARG;
    try {
        h->ns = ns ? ns : "";
    } catch (...) {
        return MIXQ_E_BADARG;
    }
    return MIXQ_OK;
}

const char* mixq_get_namespace(const mixq_handle* h) { return h ? h->ns.c_str() : ""; }

// --------------------------------------------------------------------------------- shape negotiation ---
int mixq_get_nb_outputs(const mixq_handle*) { return 1; }

int mixq_get_output_dimensions(const mixq_handle*, int outputIndex, const mixq_tensor_desc* inputs, int nbInputs,
                               mixq_tensor_desc* out)
{
    if (outputIndex != 0 || !inputs || !out || nbInputs < 2) return MIXQ_E_BADARG;
    const int nb = inputs[0].nbDims;
    if (nb < 1 || nb > MIXQ_MAX_DIMS || inputs[1].nbDims < 1) return MIXQ_E_BADARG;
    *out = inputs[0];
    out->d[nb - 1] = inputs[1].d[0];
    return MIXQ_OK;
}

int mixq_supports_format_combination(const mixq_handle*, int pos, const mixq_tensor_desc* inOut, int nbInputs,
                                     int nbOutputs)
{
    if (!inOut || pos < 0 || pos >= nbInputs + nbOutputs || pos > 7) return 0;
    return (inOut[pos].type == MIXQ_TYPE_HALF && inOut[pos].format == MIXQ_FORMAT_LINEAR) ? 1 : 0;
}

int mixq_get_output_data_type(const mixq_handle*, int) { return MIXQ_TYPE_HALF; }

// ------------------------------------------------------------------------------------------ workspace ---
// Bytes of the qA region of the workspace carve for a call with M rows: M x K row-major -- or, for decode batches, room for the
// skinny GEMM's fragment-major image (whole 16-row tiles x whole 64-byte k-steps: quant_kernels.hip FRAG), whichever is larger.
static size_t qa_region_bytes(int64_t M, int64_t K)
{
    size_t b = (size_t)M * (size_t)K;
    if (M > kSmallMFastPath && M <= 64) {
        const size_t f = (size_t)((M + 15) / 16 * 16) * (size_t)((K + 63) / 64 * 64);
        if (f > b) b = f;
    }
    return b;
}

// Exchange scratch ONE mixq_enqueue call with exactly M rows carves behind fpA (the K splits over workgroups of
// gemm_pp_kernels.hip / gemm_kernels.hip): the 256x256 form's from 129 rows on where its plan applies, else the small-tile form's.
// enqueue_impl and the workspace bound below both go through this function, so they cannot disagree.
static size_t enqueue_scratch_bytes(int64_t M, int64_t N, int64_t K)
{
    if (M <= 4 || N <= 0 || K <= 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return 0;
    // (the mid-M deep form is chosen first where its table says so; the scratch also covers whichever form a forced schedule would
    //  fall back to on the same shape)
    const size_t d = mixq::gemm_deep_takes((int)M, (int)N, (int)K, true) ? mixq::gemm_deep_workspace_size((int)M, (int)N, (int)K) : 0;
    size_t a = M > 128 ? mixq::gemm_splitk_workspace_size((int)M, (int)N, (int)K) : 0;
    if (!a) a = mixq::gemm_xsplit_workspace_size((int)M, (int)N, (int)K);
    return d > a ? d : a;
}
// Largest exchange scratch mixq_enqueue can carve with these N, K and ANY M <= maxM.  The plans (gemm_splitk_plan, xsplit_plan,
// deep_plan_auto) see M mostly through ceil(M / 32 | 64 | 128 | 256) and the thresholds 4 / 16 / 32 / 128 / 256, and are not monotone
// in M, so every 32-row step is probed at its right end (where those ceilings are constant over the step) plus the thresholds --
// and, because gemm_pp128_wins weighs 129..256 rows by M itself, EVERY M of that range; N <= 0 or a huge maxM: the
// shape-independent bound.
static size_t enqueue_scratch_bound(int64_t maxM, int64_t N, int64_t K)
{
    if (maxM <= 4) return 0;
    const size_t any = mixq::gemm_splitk_workspace_bound() > mixq::gemm_xsplit_workspace_bound()
                           ? mixq::gemm_splitk_workspace_bound()
                           : mixq::gemm_xsplit_workspace_bound();
    if (N <= 0 || N > INT32_MAX || K > INT32_MAX || maxM > INT32_MAX) return any;
    const int64_t steps = (maxM + 31) / 32;
    if (steps > (int64_t)1 << 20) return any;
    size_t best = 0;
    auto probe = [&](int64_t m) {
        if (m < 5 || m > maxM) return;
        const size_t b = enqueue_scratch_bytes(m, N, K);
        if (b > best) best = b;
    };
    for (int64_t m : {5, 8, 16, 17}) probe(m);
    for (int64_t m = 129; m <= 256 && m <= maxM; ++m) probe(m);
    for (int64_t t = 1; t <= steps; ++t) probe(t * 32 < maxM ? t * 32 : maxM);
    probe(maxM);
    return best;
}

size_t mixq_enqueue_scratch_size(int64_t M, int64_t N, int64_t K) { return enqueue_scratch_bytes(M, N, K); }

size_t mixq_workspace_size(const mixq_handle*, int64_t maxM, int64_t N, int64_t K)
{
    if (maxM <= 0 || K <= 0) return kWorkspaceAlign;
    size_t s = kWorkspaceAlign; // slack for aligning the base like nextWorkspacePtr(ptr, 0)
    {   // qA: the largest region any M <= maxM carves (decode batches may hold the padded fragment-major image)
        size_t q = qa_region_bytes(maxM, K);
        const size_t q64 = qa_region_bytes(maxM < 64 ? maxM : 64, K);
        s += align_up(q > q64 ? q : q64, kWorkspaceAlign);
    }
    s += align_up((size_t)maxM * sizeof(uint16_t), kWorkspaceAlign);                          // sA
    s += align_up((size_t)maxM * (size_t)kNumOutliers * sizeof(uint16_t), kWorkspaceAlign);   // fpA
    // exchange scratch of the K splits over workgroups (csrc/gemm_pp_kernels.hip, gemm_kernels.hip): only what a call
    // with this N, K and at most maxM rows can actually use -- 0 for most shapes
    const size_t scratch = enqueue_scratch_bound(maxM, N, K);
    if (scratch) s += align_up(scratch, kWorkspaceAlign);
    return s;
}

size_t mixq_reference_workspace_size(int64_t maxM, int64_t N, int64_t K)
{
    if (maxM < 0 || N < 0 || K < 0) return 0;
    const size_t a = (size_t)maxM * (size_t)K + (size_t)maxM * 2 + (size_t)K * (size_t)N * 2;
    const size_t b = (size_t)maxM * (size_t)N * 2 * 8;
    const size_t w = a > b ? a : b;
    return w ? w : (size_t)33554432; // CUBLAS_WORKSPACE_SIZE fallback, TsinghuaMixQPlugin.cpp:23,372-376
}

// ---------------------------------------------------------------------------------------- launchers ----
int mixq_int8quant(int rows, int cols, const void* src, int8_t* output, void* scale, void* stream)
{
    if (rows < 0 || cols <= 0 || (rows > 0 && (!src || !output || !scale))) return MIXQ_E_BADARG;
    if (cols % 8) return MIXQ_E_SHAPE;
    if (!aligned16(src) || (reinterpret_cast<uintptr_t>(output) & 7u)) return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_quant_extract(const_cast<void*>(src), output, scale, nullptr, nullptr, rows, cols, 0,
                                             false, static_cast<hipStream_t>(stream)));
}

int mixq_extract_outliers(int M, int K, const void* A, void* fpA, const int32_t* ind, int len, void* stream)
{
    if (M < 0 || K <= 0 || len < 0 || (M > 0 && len > 0 && (!A || !fpA || !ind))) return MIXQ_E_BADARG;
    return hip_rc(mixq::launch_extract(const_cast<void*>(A), fpA, ind, M, K, len, false,
                                       static_cast<hipStream_t>(stream)));
}

int mixq_extract_outliers_set_zero(int M, int K, void* A, void* fpA, const int32_t* ind, int len, void* stream)
{
    if (M < 0 || K <= 0 || len < 0 || (M > 0 && len > 0 && (!A || !fpA || !ind))) return MIXQ_E_BADARG;
    return hip_rc(mixq::launch_extract(A, fpA, ind, M, K, len, true, static_cast<hipStream_t>(stream)));
}

size_t mixq_find_outliers_workspace_size(int K) { return K > 0 ? (size_t)((K + 31) / 32) * 4 : 0; }

int mixq_find_outliers(const void* A, int M, int K, float sigma, void* mask_ws, int32_t* ind_out, int32_t* count_out,
                       int capacity, void* stream)
{
    if (M < 0 || K <= 0 || capacity < 0 || !mask_ws || !count_out || (capacity > 0 && !ind_out) || (M > 0 && !A))
        return MIXQ_E_BADARG;
    if (K % 8) return MIXQ_E_SHAPE;
    if (M > 0 && !aligned16(A)) return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_find_outliers(A, M, K, sigma, static_cast<unsigned*>(mask_ws), ind_out, count_out,
                                             capacity, static_cast<hipStream_t>(stream)));
}

int mixq_dequant_weight_columns(const int8_t* q_weight, const void* scale_col, const int32_t* ind, int len, void* out,
                                int N, int K, void* stream)
{
    if (N < 0 || K <= 0 || len < 0 || (N > 0 && len > 0 && (!q_weight || !scale_col || !ind || !out)))
        return MIXQ_E_BADARG;
    return hip_rc(mixq::launch_dequant_columns(q_weight, scale_col, ind, len, out, N, K,
                                               static_cast<hipStream_t>(stream)));
}

int mixq_quant_extract(int M, int K, void* A, int8_t* qA, void* sA, void* fpA, const int32_t* ind, int len,
                       int zero_outliers, void* stream)
{
    if (M < 0 || K <= 0 || len < 0 || (M > 0 && (!A || !qA || !sA))) return MIXQ_E_BADARG;
    if (len > 0 && (!fpA || !ind)) return MIXQ_E_BADARG;
    if (K % 8) return MIXQ_E_SHAPE;
    if (!aligned16(A) || (reinterpret_cast<uintptr_t>(qA) & 7u)) return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_quant_extract(A, qA, sA, len > 0 ? fpA : nullptr, ind, M, K, len, zero_outliers != 0,
                                             static_cast<hipStream_t>(stream)));
}

int mixq_int8_quantize_with_scale(int rows, int cols, const void* src, const void* scale, int8_t* output, void* stream)
{
    if (rows < 0 || cols <= 0 || (rows > 0 && (!src || !scale || !output))) return MIXQ_E_BADARG;
    if (cols % 8) return MIXQ_E_SHAPE;
    if (!aligned16(src) || (reinterpret_cast<uintptr_t>(output) & 7u)) return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_quant_with_scale(src, scale, output, rows, cols, static_cast<hipStream_t>(stream)));
}

int mixq_rmsnorm(int M, int K, const void* x, const void* gamma, void* out, float eps, void* stream)
{
    if (M < 0 || K <= 0 || (M > 0 && (!x || !gamma || !out))) return MIXQ_E_BADARG;
    if (K % 8 || K > 32768) return MIXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(gamma) || !aligned16(out)) return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_rmsnorm_quant(x, gamma, out, nullptr, nullptr, nullptr, nullptr, eps, M, K, 0, false,
                                             static_cast<hipStream_t>(stream)));
}

int mixq_rmsnorm_extract_quant(int M, int K, const void* x, const void* gamma, void* out, float eps, const int32_t* ind,
                               int len, void* outliers, int8_t* q, void* scale, void* stream)
{
    if (M < 0 || K <= 0 || len < 0 || (M > 0 && (!x || !gamma || !out || !q || !scale))) return MIXQ_E_BADARG;
    if (len > 0 && (!ind || !outliers)) return MIXQ_E_BADARG;
    if (K % 8 || K > 32768) return MIXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(gamma) || !aligned16(out) || (reinterpret_cast<uintptr_t>(q) & 7u)) return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_rmsnorm_quant(x, gamma, out, outliers, ind, q, scale, eps, M, K, len, 8,
                                             static_cast<hipStream_t>(stream)));
}

int mixq_rmsnorm_extract_quant4(int M, int K, const void* x, const void* gamma, void* out, float eps, const int32_t* ind,
                                int len, void* outliers, uint8_t* q_packed, void* scale, void* stream)
{
    if (M < 0 || K <= 0 || len < 0 || (M > 0 && (!x || !gamma || !out || !q_packed || !scale))) return MIXQ_E_BADARG;
    if (len > 0 && (!ind || !outliers)) return MIXQ_E_BADARG;
    if (K % 8 || K > 32768) return MIXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(gamma) || !aligned16(out) || (reinterpret_cast<uintptr_t>(q_packed) & 3u))
        return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_rmsnorm_quant(x, gamma, out, outliers, ind, reinterpret_cast<int8_t*>(q_packed), scale, eps,
                                             M, K, len, 4, static_cast<hipStream_t>(stream)));
}

static int fused_dequant_impl(const int8_t* A, const int8_t* B, const void* scale_row, const void* scale_col,
                              const void* y, void* D, int M, int N, int K, int epi, void* stream,
                              const void* mul = nullptr, void* scratch = nullptr, int a_frag = 0)
{
    if (M < 0 || N < 0 || K <= 0) return MIXQ_E_BADARG;
    if (M == 0 || N == 0) return MIXQ_OK;
    if (!A || !B || !scale_row || !scale_col || !D) return MIXQ_E_BADARG;
    if (K % 16 || N % 16) return MIXQ_E_SHAPE; // CUTLASS 16-B alignment of the reference
    if (!aligned16(A) || !aligned16(B) || !aligned16(D) || !aligned16(scale_col) || (y && !aligned16(y)))
        return MIXQ_E_ALIGN;
    mixq::GemmParams p{};
    p.A = A, p.B = B;
    p.sA = static_cast<const uint16_t*>(scale_row), p.sW = static_cast<const uint16_t*>(scale_col);
    p.Y = static_cast<const uint16_t*>(y), p.D = D;
    p.Mul = static_cast<const uint16_t*>(mul);
    if (epi == mixq::EPI_DEQUANT_SILU_MUL && (!mul || (reinterpret_cast<uintptr_t>(mul) & 7u))) return MIXQ_E_BADARG;
    p.zeros = mixq::zero_page();
    p.M = M, p.N = N, p.K = K, p.O = 0;
    if (!p.zeros) return MIXQ_E_HIP;
    if (scratch && aligned16(scratch)) p.splitk_ws = scratch; // K split over workgroups where the shape calls for it
    p.a_frag = a_frag;
    if (M <= 64) p.b_image = mixq::resolve_weight_image(B, N, K, static_cast<hipStream_t>(stream)); // (ONE lookup per call)
    if (a_frag != 0 && !(a_frag == 1 && mixq::gemm_takes_skinny(p, epi))) return MIXQ_E_SHAPE; // (the image has ONE reader)
    return hip_rc(mixq::launch_gemm(p, epi, static_cast<hipStream_t>(stream)));
}

int mixq_int8_fused_dequantize(const int8_t* A, const int8_t* B, const void* scale_row, const void* scale_col,
                               const void* y, void* D, int M, int N, int K, char* workspace, void* stream)
{
    return fused_dequant_impl(A, B, scale_row, scale_col, y, D, M, N, K, mixq::EPI_DEQUANT, stream, nullptr, workspace);
}

int mixq_int8_fused_dequantize_silu(const int8_t* A, const int8_t* B, const void* scale_row, const void* scale_col,
                                    const void* y, void* D, int M, int N, int K, char* workspace, void* stream)
{
    return fused_dequant_impl(A, B, scale_row, scale_col, y, D, M, N, K, mixq::EPI_DEQUANT_SILU, stream, nullptr,
                              workspace);
}

int mixq_int4quant(int rows, int cols, const void* src, uint8_t* dst, void* scale, void* stream)
{
    if (rows < 0 || cols <= 0 || (rows > 0 && (!src || !dst || !scale))) return MIXQ_E_BADARG;
    if (cols % 8) return MIXQ_E_SHAPE;
    if (rows > 0 && (!aligned16(src) || (reinterpret_cast<uintptr_t>(dst) & 3u))) return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_quant4_rows(src, dst, scale, rows, cols, static_cast<hipStream_t>(stream)));
}

static size_t align16_up(size_t v) { return (v + 15) & ~static_cast<size_t>(15); }
static std::atomic<int> g_int4_stream{1};
static void set_int4_stream(int on) { g_int4_stream.store(on); }
static std::atomic<int> g_int4_fuse_quant{1}; // test / measurement knob 875 (default: by rule) / 876 (never) / 877 (always)
static void set_int4_fuse_quant(int on) { g_int4_fuse_quant.store(on); }
// ^ test / measurement knob 870 (default: on) / 871: decode batches through the unpack route

size_t mixq_int4_fused_workspace_size(int M, int N, int k_packed)
{
    if (M < 0 || N < 0 || k_packed < 0) return 0;
    // (round 6, ADVICE r5: the size the call REALLY needs -- 0 where mixq_int4_fused_dequantize[_silu] streams the packed weight (decode batches
    //  that int4_gemm_kernels.hip serves, knob 870 on); N == 0 asks for the activation's part alone, the _w8 entry's need)
    if (N > 0 && g_int4_stream.load(std::memory_order_relaxed) && mixq::gemm_skinny_s4_supported(M, N, k_packed)) return 0;
    return align16_up((size_t)M * 2 * k_packed) + align16_up((size_t)N * 2 * k_packed);
}

// b8 != nullptr: the weight already widened to int8 [N, 2 k_packed] (once, at load time); only A is unpacked per call
static int int4_fused_impl(const uint8_t* A, const uint8_t* B, const void* scale_row, const void* scale_col,
                           const void* y, void* D, int M, int N, int k_packed, char* workspace, int epi, void* stream,
                           const int8_t* b8 = nullptr)
{
    if (M < 0 || N < 0 || k_packed <= 0) return MIXQ_E_BADARG;
    if (M == 0 || N == 0) return MIXQ_OK;
    if (!A || (!B && !b8)) return MIXQ_E_BADARG;
    if (k_packed % 16) return MIXQ_E_SHAPE;
    if (!aligned16(A) || (B && !aligned16(B)) || (b8 && !aligned16(b8))) return MIXQ_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // decode batches: the packed weight stream (int4_gemm_kernels.hip) -- ONE launch, no workspace, N K / 2 weight bytes
    if (B && g_int4_stream.load(std::memory_order_relaxed) && mixq::gemm_skinny_s4_supported(M, N, k_packed)) {
        if (!scale_row || !scale_col || !D) return MIXQ_E_BADARG;
        if (!aligned16(D) || !aligned16(scale_col) || (y && !aligned16(y))) return MIXQ_E_ALIGN;
        mixq::GemmParams p{};
        p.A = reinterpret_cast<const int8_t*>(A), p.B = reinterpret_cast<const int8_t*>(B);
        p.sA = static_cast<const uint16_t*>(scale_row), p.sW = static_cast<const uint16_t*>(scale_col);
        p.Y = static_cast<const uint16_t*>(y), p.D = D;
        p.M = M, p.N = N, p.K = k_packed, p.O = 0;
        return hip_rc(mixq::launch_gemm_skinny_s4(p, epi, st));
    }
    if (!workspace) return MIXQ_E_WORKSPACE;
    if (!aligned16(workspace)) return MIXQ_E_ALIGN;
    int8_t* a8 = reinterpret_cast<int8_t*>(workspace);
    hipError_t e = mixq::launch_unpack_s4(A, a8, (size_t)M * k_packed, st);
    if (e == hipSuccess && !b8) {
        int8_t* w8 = a8 + align16_up((size_t)M * 2 * k_packed);
        e = mixq::launch_unpack_s4(B, w8, (size_t)N * k_packed, st);
        b8 = w8;
    }
    if (e != hipSuccess) return hip_rc(e);
    return fused_dequant_impl(a8, b8, scale_row, scale_col, y, D, M, N, 2 * k_packed, epi, stream);
}

// One call for the 4-bit flavour's linear on an fp16 activation (round 6): FindRowScale(bit = 4) + int4FusedDequantize[Silu]
// (cult.cu:2515-2567 + 2005-2060 / 2119-2181).  ONE row runs as ONE launch (the row is quantised inside the weight-streaming kernel: -5..-12 %
// against two launches); from two rows on the in-kernel quantiser loses to the quantiser launch it replaces (int4_gemm_kernels.hip, QF) and
// the call is the two launches, through `q_packed`.
int mixq_int4_linear_forward(const void* x, const uint8_t* B, void* x_scale, uint8_t* q_packed, const void* scale_col, const void* y,
                             void* D, int M, int N, int k_packed, int epilogue, char* workspace, void* stream)
{
    if (epilogue != mixq::EPI_DEQUANT && epilogue != mixq::EPI_DEQUANT_SILU) return MIXQ_E_BADARG;
    if (M < 0 || N < 0 || k_packed <= 0) return MIXQ_E_BADARG;
    if (M == 0 || N == 0) return MIXQ_OK;
    if (!x || !B || !x_scale || !scale_col || !D) return MIXQ_E_BADARG;
    if (k_packed % 16) return MIXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(B) || !aligned16(D) || !aligned16(scale_col) || (y && !aligned16(y))) return MIXQ_E_ALIGN;
    const int fq = g_int4_fuse_quant.load(std::memory_order_relaxed);
    if (g_int4_stream.load(std::memory_order_relaxed) && (fq == 2 || (fq == 1 && M == 1)) && mixq::gemm_skinny_s4q_supported(M, N, k_packed)) {
        mixq::GemmParams p{};
        p.A = static_cast<const int8_t*>(x), p.B = reinterpret_cast<const int8_t*>(B);
        p.sA = static_cast<const uint16_t*>(x_scale), p.sW = static_cast<const uint16_t*>(scale_col);
        p.Y = static_cast<const uint16_t*>(y), p.D = D;
        p.M = M, p.N = N, p.K = k_packed, p.O = 0;
        return hip_rc(mixq::launch_gemm_skinny_s4q(p, epilogue, static_cast<hipStream_t>(stream)));
    }
    if (!q_packed) return MIXQ_E_WORKSPACE;
    const int rc = mixq_int4quant(M, 2 * k_packed, x, q_packed, x_scale, stream);
    if (rc != MIXQ_OK) return rc;
    return int4_fused_impl(q_packed, B, x_scale, scale_col, y, D, M, N, k_packed, workspace, epilogue, stream);
}

int mixq_int4_fused_dequantize_w8(const uint8_t* A, const int8_t* B_int8, const void* scale_row, const void* scale_col,
                                  const void* y, void* D, int M, int N, int k_packed, int epilogue, char* workspace, void* stream)
{
    if (epilogue != mixq::EPI_DEQUANT && epilogue != mixq::EPI_DEQUANT_SILU) return MIXQ_E_BADARG;
    if (!B_int8) return MIXQ_E_BADARG;
    return int4_fused_impl(A, nullptr, scale_row, scale_col, y, D, M, N, k_packed, workspace, epilogue, stream, B_int8);
}

int mixq_int4_fused_dequantize(const uint8_t* A, const uint8_t* B, const void* scale_row, const void* scale_col,
                               const void* y, void* D, int M, int N, int k_packed, char* workspace, void* stream)
{
    return int4_fused_impl(A, B, scale_row, scale_col, y, D, M, N, k_packed, workspace, mixq::EPI_DEQUANT, stream);
}

int mixq_int4_fused_dequantize_silu(const uint8_t* A, const uint8_t* B, const void* scale_row, const void* scale_col,
                                    const void* y, void* D, int M, int N, int k_packed, char* workspace, void* stream)
{
    return int4_fused_impl(A, B, scale_row, scale_col, y, D, M, N, k_packed, workspace, mixq::EPI_DEQUANT_SILU, stream);
}

int mixq_unpack_int4_to_fp16(const uint8_t* weight, const int32_t* ind, int rows, int cols_packed, int n, void* out,
                             void* stream)
{
    if (rows < 0 || cols_packed <= 0 || n < 0 || (rows > 0 && n > 0 && (!weight || !ind || !out))) return MIXQ_E_BADARG;
    return hip_rc(mixq::launch_unpack_s4_columns(weight, ind, rows, cols_packed, n, out,
                                                 static_cast<hipStream_t>(stream)));
}

int mixq_unpack_int4_to_int8(const uint8_t* src, int8_t* dst, size_t packed_bytes, void* stream)
{
    if (packed_bytes == 0) return MIXQ_OK;
    if (!src || !dst) return MIXQ_E_BADARG;
    if (packed_bytes % 16) return MIXQ_E_SHAPE;
    if (!aligned16(src) || !aligned16(dst)) return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_unpack_s4(src, dst, packed_bytes, static_cast<hipStream_t>(stream)));
}

int mixq_int8_fused_dequantize_silu_mul(const int8_t* A, const int8_t* B, const void* scale_row, const void* scale_col,
                                        const void* y, const void* mul, void* D, int M, int N, int K,
                                        char* workspace, void* stream)
{
    return fused_dequant_impl(A, B, scale_row, scale_col, y, D, M, N, K, mixq::EPI_DEQUANT_SILU_MUL, stream, mul,
                              workspace);
}

// scratch of whichever cross-workgroup K split launch_gemm would pick for this shape (at most one applies)
static size_t gemm_scratch_bytes(int M, int N, int K)
{
    const size_t d = mixq::gemm_deep_takes(M, N, K, true) ? mixq::gemm_deep_workspace_size(M, N, K) : 0;
    size_t a = mixq::gemm_splitk_workspace_size(M, N, K);
    if (!a) a = mixq::gemm_xsplit_workspace_size(M, N, K);
    return d > a ? d : a;
}

size_t mixq_gemm_scratch_size(int M, int N, int K) { return gemm_scratch_bytes(M, N, K); }

size_t mixq_gemm_scratch_bound(void)
{
    const size_t a = mixq::gemm_splitk_workspace_bound(), b = mixq::gemm_xsplit_workspace_bound();
    return a > b ? a : b;
}

int mixq_gemm_mixed(const int8_t* qA, const int8_t* W, const void* sA, const void* sW, const void* fpA,
                    const void* fpW, void* Out, int M, int N, int K, int O, void* stream)
{
    return mixq_gemm_mixed_scratch(qA, W, sA, sW, fpA, fpW, Out, M, N, K, O, nullptr, 0, stream);
}

static int gemm_mixed_impl(const int8_t* qA, const int8_t* W, const void* sA, const void* sW, const void* fpA, const void* fpW,
                           void* Out, int M, int N, int K, int O, void* scratch, size_t scratch_bytes, void* stream,
                           int a_frag, const void* b_image = nullptr, bool image_resolved = false);

int mixq_gemm_mixed_scratch(const int8_t* qA, const int8_t* W, const void* sA, const void* sW, const void* fpA,
                            const void* fpW, void* Out, int M, int N, int K, int O, void* scratch, size_t scratch_bytes,
                            void* stream)
{
    return gemm_mixed_impl(qA, W, sA, sW, fpA, fpW, Out, M, N, K, O, scratch, scratch_bytes, stream, 0);
}

// a_frag: layout of qA (GemmParams::a_frag): enqueue's own quantiser may write the consuming kernel's preferred image; the
// public entries take row-major qA
// b_image / image_resolved: the caller has already looked the weight's image up for this call (one lookup per call)
static int gemm_mixed_impl(const int8_t* qA, const int8_t* W, const void* sA, const void* sW, const void* fpA, const void* fpW,
                           void* Out, int M, int N, int K, int O, void* scratch, size_t scratch_bytes, void* stream,
                           int a_frag, const void* b_image, bool image_resolved)
{
    if (M < 0 || N < 0 || K <= 0 || O < 0) return MIXQ_E_BADARG;
    if (M == 0 || N == 0) return MIXQ_OK;
    if (!qA || !W || !sA || !sW || !Out || (O > 0 && (!fpA || !fpW))) return MIXQ_E_BADARG;
    if (K % 16 || N % 16 || O % 8) return MIXQ_E_SHAPE;
    if (!aligned16(qA) || !aligned16(W) || !aligned16(Out) || !aligned16(sW) || (O > 0 && (!aligned16(fpA) || !aligned16(fpW))))
        return MIXQ_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mixq::GemmParams p{};
    p.A = qA, p.B = W;
    p.sA = static_cast<const uint16_t*>(sA), p.sW = static_cast<const uint16_t*>(sW);
    p.D = Out;
    p.zeros = mixq::zero_page();
    if (!p.zeros) return MIXQ_E_HIP;
    p.M = M, p.N = N, p.K = K;
    p.dbg = g_dbg_stamps.load(std::memory_order_relaxed);
    p.a_frag = a_frag;
    p.b_image = image_resolved ? b_image : (M <= 64 ? mixq::resolve_weight_image(W, N, K, st) : nullptr);
    if (scratch && aligned16(scratch) && scratch_bytes >= gemm_scratch_bytes(M, N, K)) p.splitk_ws = scratch;
    if (O <= kNumOutliers) {
        p.fpA = static_cast<const uint16_t*>(fpA), p.fpW = static_cast<const uint16_t*>(fpW), p.O = O;
        return hip_rc(mixq::launch_gemm(p, mixq::EPI_DEQUANT, st));
    }
    // more outlier columns than one LDS pass holds: the reference's own two-step order (side GEMM into Out, then C = D = Out)
    hipError_t e = mixq::launch_gemm_fp16(fpA, fpW, Out, M, N, O, st);
    if (e != hipSuccess) return MIXQ_E_HIP;
    p.Y = static_cast<const uint16_t*>(Out), p.O = 0;
    return hip_rc(mixq::launch_gemm(p, mixq::EPI_DEQUANT, st));
}

// ---- weight images (MI355X extension): a fragment-major copy of `weight` for the decode-batch GEMM ------------------------------
size_t mixq_weight_image_bytes(int64_t N, int64_t K)
{
    if (N <= 0 || K <= 0 || N % 16 || K % 64 || N > INT32_MAX || K > INT32_MAX) return 0;
    return (size_t)N * (size_t)K;
}

int mixq_weight_image_register(const int8_t* weight, int64_t N, int64_t K, void* image, void* stream)
{
    if (!weight || !image) return MIXQ_E_BADARG;
    if (mixq_weight_image_bytes(N, K) == 0) return MIXQ_E_SHAPE;
    if (!aligned16(weight) || !aligned16(image)) return MIXQ_E_ALIGN;
    int rc = hip_rc(mixq::launch_weight_image(weight, static_cast<int8_t*>(image), (int)N, (int)K, static_cast<hipStream_t>(stream)));
    if (rc == MIXQ_OK) rc = hip_rc(mixq::register_weight_image(weight, image, (int)N, (int)K, static_cast<hipStream_t>(stream)));
    return rc;
}

int mixq_weight_image_unregister(const int8_t* weight) { return weight && mixq::unregister_weight_image(weight) ? MIXQ_OK : MIXQ_E_BADARG; }

int mixq_weight_image_verify(const int8_t* weight, void* stream)
{
    if (!weight) return MIXQ_E_BADARG;
    const int v = mixq::verify_weight_image(weight, static_cast<hipStream_t>(stream));
    return v == 1 ? MIXQ_OK : v == 0 ? MIXQ_E_STALE : MIXQ_E_BADARG;
}

int mixq_weight_image_stale_count(void) { return mixq::weight_image_stale_count(); }

// ---- qA layouts (MI355X extension): the producer may write the image its consumer reads fastest ---------------------------
int mixq_qa_layout(int M, int N, int K)
{
    if (M <= kSmallMFastPath || M > 64 || N <= 0 || K <= 0 || K % 16 || N % 16) return MIXQ_QA_ROW_MAJOR;
    if (!mixq::qa_frag_enabled() || !mixq::quant_frag_layout_supported(M, K)) return MIXQ_QA_ROW_MAJOR;
    mixq::GemmParams probe{};
    probe.M = M, probe.N = N, probe.K = K, probe.O = kNumOutliers, probe.a_frag = 1;
    static int dummy;
    if (gemm_scratch_bytes(M, N, K) != 0) probe.splitk_ws = &dummy; // (where a caller WITH scratch would get a K split over workgroups
                                                                    //  instead of the skinny kernel, the answer must be row-major)
    return mixq::gemm_takes_skinny(probe, mixq::EPI_DEQUANT) ? MIXQ_QA_FRAGMENT_MAJOR : MIXQ_QA_ROW_MAJOR;
}

size_t mixq_qa_bytes(int M, int K, int layout)
{
    if (M <= 0 || K <= 0) return 0;
    if (layout == MIXQ_QA_FRAGMENT_MAJOR) return (size_t)((M + 15) / 16 * 16) * (size_t)((K + 63) / 64 * 64);
    return (size_t)M * (size_t)K;
}

int mixq_quant_extract_layout(int M, int K, void* A, int8_t* qA, void* sA, void* fpA, const int32_t* ind, int len,
                              int zero_outliers, int q_layout, void* stream)
{
    if (q_layout == MIXQ_QA_ROW_MAJOR) return mixq_quant_extract(M, K, A, qA, sA, fpA, ind, len, zero_outliers, stream);
    if (q_layout != MIXQ_QA_FRAGMENT_MAJOR) return MIXQ_E_BADARG;
    if (M < 0 || K <= 0 || len < 0 || (M > 0 && (!A || !qA || !sA))) return MIXQ_E_BADARG;
    if (len > 0 && (!fpA || !ind)) return MIXQ_E_BADARG;
    if (K % 8 || !mixq::quant_frag_layout_supported(M, K)) return MIXQ_E_SHAPE;
    if (!aligned16(A) || !aligned16(qA)) return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_quant_extract(A, qA, sA, len > 0 ? fpA : nullptr, ind, M, K, len, zero_outliers != 0,
                                             static_cast<hipStream_t>(stream), nullptr, 1));
}

int mixq_rmsnorm_extract_quant_layout(int M, int K, const void* x, const void* gamma, void* out, float eps, const int32_t* ind,
                                      int len, void* outliers, int8_t* q, void* scale, int q_layout, void* stream)
{
    if (q_layout == MIXQ_QA_ROW_MAJOR) return mixq_rmsnorm_extract_quant(M, K, x, gamma, out, eps, ind, len, outliers, q, scale, stream);
    if (q_layout != MIXQ_QA_FRAGMENT_MAJOR) return MIXQ_E_BADARG;
    if (M < 0 || K <= 0 || len < 0 || (M > 0 && (!x || !gamma || !out || !q || !scale))) return MIXQ_E_BADARG;
    if (len > 0 && (!ind || !outliers)) return MIXQ_E_BADARG;
    if (K % 8 || K > 32768 || !mixq::quant_frag_layout_supported(M, K)) return MIXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(gamma) || !aligned16(out) || !aligned16(q)) return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_rmsnorm_quant(x, gamma, out, outliers, ind, q, scale, eps, M, K, len, 8,
                                             static_cast<hipStream_t>(stream), 1));
}

int mixq_int8_fused_dequantize_layout(const int8_t* A, const int8_t* B, const void* scale_row, const void* scale_col,
                                      const void* y, const void* mul, void* D, int M, int N, int K, int epilogue, int qa_layout,
                                      char* workspace, void* stream)
{
    if (epilogue != mixq::EPI_DEQUANT && epilogue != mixq::EPI_DEQUANT_SILU && epilogue != mixq::EPI_DEQUANT_SILU_MUL)
        return MIXQ_E_BADARG;
    if (qa_layout != MIXQ_QA_ROW_MAJOR && qa_layout != MIXQ_QA_FRAGMENT_MAJOR) return MIXQ_E_BADARG;
    return fused_dequant_impl(A, B, scale_row, scale_col, y, D, M, N, K, epilogue, stream, mul, workspace, qa_layout);
}

int mixq_gemm_mixed_layout(const int8_t* qA, const int8_t* W, const void* sA, const void* sW, const void* fpA, const void* fpW,
                           void* Out, int M, int N, int K, int O, int qa_layout, void* scratch, size_t scratch_bytes, void* stream)
{
    if (qa_layout != MIXQ_QA_ROW_MAJOR && qa_layout != MIXQ_QA_FRAGMENT_MAJOR) return MIXQ_E_BADARG;
    const void* img = nullptr;
    bool resolved = false;
    if (qa_layout == MIXQ_QA_FRAGMENT_MAJOR) {
        if (O > kNumOutliers || M <= 0 || N <= 0 || K <= 0 || !W) return MIXQ_E_SHAPE;
        mixq::GemmParams probe{};
        probe.M = M, probe.N = N, probe.K = K, probe.O = O, probe.a_frag = 1, probe.B = W;
        probe.b_image = img = mixq::resolve_weight_image(W, N, K, static_cast<hipStream_t>(stream));
        resolved = true;
        probe.splitk_ws = (scratch && scratch_bytes >= gemm_scratch_bytes(M, N, K) && gemm_scratch_bytes(M, N, K)) ? scratch : nullptr;
        if (!mixq::gemm_takes_skinny(probe, mixq::EPI_DEQUANT)) return MIXQ_E_SHAPE; // (the image has ONE reader)
    }
    return gemm_mixed_impl(qA, W, sA, sW, fpA, fpW, Out, M, N, K, O, scratch, scratch_bytes, stream, qa_layout, img, resolved);
}

int mixq_mixlinear_forward(int M, int N, int K, int O, void* x, const int32_t* ind, const int8_t* q_weight,
                           const void* scale_col, const void* weight_cache, void* x_scale, int8_t* q_x, void* outliers,
                           void* out, int q_layout, void* scratch, size_t scratch_bytes, void* stream)
{
    if (M < 0 || N < 0 || K <= 0 || O < 0) return MIXQ_E_BADARG;
    if (M == 0 || N == 0) return MIXQ_OK;
    if (!x || !q_weight || !scale_col || !x_scale || !q_x || !out || (O > 0 && (!ind || !weight_cache || !outliers)))
        return MIXQ_E_BADARG;
    // two launches instead of the four of MixLinear_GEMM.forward (linear.py:163-286): ExtractOutliersAndSetToZeros +
    // FindRowScale = the fused producer with zero_outliers = 1 (cult.cu:2616-2709 is the reference's own fused form), the
    // outlier product + int8FusedDequantize = the fused GEMM with its fp16 side product (<= 128 outlier columns, a multiple
    // of 8; anything else takes the reference's own two steps inside mixq_gemm_mixed_scratch)
    // q_layout: MIXQ_QA_FRAGMENT_MAJOR only where mixq_qa_layout(M, N, K) says so (q_x then holds mixq_qa_bytes(M, K, 1) bytes and
    // is opaque to the caller; later consumers of it -- the gate projection -- must pass the same layout)
    if (q_layout == MIXQ_QA_FRAGMENT_MAJOR && (O % 8 != 0 || O > kNumOutliers || mixq_qa_layout(M, N, K) != MIXQ_QA_FRAGMENT_MAJOR))
        return MIXQ_E_SHAPE;
    int rc = mixq_quant_extract_layout(M, K, x, q_x, x_scale, outliers, ind, O, 1, q_layout, stream);
    if (rc != MIXQ_OK) return rc;
    if (O % 8 == 0)
        return mixq_gemm_mixed_layout(q_x, q_weight, x_scale, scale_col, outliers, weight_cache, out, M, N, K, O, q_layout,
                                      q_layout ? nullptr : scratch, q_layout ? 0 : scratch_bytes, stream);
    rc = mixq_gemm_fp16(outliers, weight_cache, out, M, N, O, stream);
    if (rc != MIXQ_OK) return rc;
    return mixq_int8_fused_dequantize(q_x, q_weight, x_scale, scale_col, out, out, M, N, K,
                                      scratch_bytes >= mixq_gemm_scratch_size(M, N, K) ? static_cast<char*>(scratch) : nullptr,
                                      stream);
}

int mixq_gemm_s8s8s32(const int8_t* A, const int8_t* B, int32_t* C, int M, int N, int K, void* stream)
{
    if (M < 0 || N < 0 || K <= 0) return MIXQ_E_BADARG;
    if (M == 0 || N == 0) return MIXQ_OK;
    if (!A || !B || !C) return MIXQ_E_BADARG;
    if (K % 16 || N % 16) return MIXQ_E_SHAPE;
    if (!aligned16(A) || !aligned16(B) || !aligned16(C)) return MIXQ_E_ALIGN;
    mixq::GemmParams p{};
    p.A = A, p.B = B, p.D = C;
    p.zeros = mixq::zero_page();
    if (!p.zeros) return MIXQ_E_HIP;
    p.M = M, p.N = N, p.K = K, p.O = 0;
    return hip_rc(mixq::launch_gemm(p, mixq::EPI_INT32, static_cast<hipStream_t>(stream)));
}

int mixq_gemm_fp16(const void* fpA, const void* fpW, void* Out, int M, int N, int O, void* stream)
{
    if (M < 0 || N < 0 || O < 0) return MIXQ_E_BADARG;
    if (M == 0 || N == 0) return MIXQ_OK;
    if (!fpA || !fpW || !Out) return MIXQ_E_BADARG;
    if (N % 4) return MIXQ_E_SHAPE; // any O: rows of odd length take the element-load path
    if ((O % 8 == 0 && (!aligned16(fpA) || !aligned16(fpW))) || (reinterpret_cast<uintptr_t>(Out) & 7u))
        return MIXQ_E_ALIGN;
    return hip_rc(mixq::launch_gemm_fp16(fpA, fpW, Out, M, N, O, static_cast<hipStream_t>(stream)));
}

int mixq_dequantization(void* out, const int32_t* x, const void* scaleRow, const void* scaleCol, int M, int N,
                        void* stream)
{
    if (M < 0 || N < 0) return MIXQ_E_BADARG;
    if (M == 0 || N == 0) return MIXQ_OK;
    if (!out || !x || !scaleRow || !scaleCol) return MIXQ_E_BADARG;
    return hip_rc(mixq::launch_dequantization(out, x, scaleRow, scaleCol, M, N, static_cast<hipStream_t>(stream)));
}

int mixq_dequantization_silu(void* out, const int32_t* x, const void* scaleRow, const void* scaleCol, const void* y, int M,
                             int N, void* stream)
{
    if (M < 0 || N < 0) return MIXQ_E_BADARG;
    if (M == 0 || N == 0) return MIXQ_OK;
    if (!out || !x || !scaleRow || !scaleCol || !y) return MIXQ_E_BADARG;
    return hip_rc(mixq::launch_dequantization_silu(out, x, scaleRow, scaleCol, y, M, N,
                                                   static_cast<hipStream_t>(stream)));
}

size_t mixq_w8a16_gemm_workspace_size(int m, int n, int k)
{
    if (m <= kSmallMFastPath || n <= 0 || k <= 0) return 0;
    return mixq::w8a16_gemm_workspace_size(m, n, k);
}

int mixq_w8a16_gemm_forward_ws(const void* input, const uint8_t* weight, const void* scale, void* output, int m, int n,
                               int k, void* workspace, size_t workspace_bytes, void* stream)
{
    if (m < 0 || n < 0 || k <= 0) return MIXQ_E_BADARG;
    if (m == 0 || n == 0) return MIXQ_OK;
    if (!input || !weight || !scale || !output) return MIXQ_E_BADARG;
    if (k % 64 || n % 2) return MIXQ_E_SHAPE; // the interleaved layout needs 64-row tiles and column pairs
    if (!aligned16(input) || !aligned16(weight)) return MIXQ_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // fpA_intB_gemm_wrapper.cu:45-70: m <= SMALL_M_FAST_PATH -> the batched GEMV, else the mixed-input tensor-core GEMM
    if (m <= kSmallMFastPath && !mixq::w8a16_skinny_takes(m, n, k))
        return hip_rc(mixq::launch_w8a16(input, weight, scale, output, m, n, k, st));
    if (k % 8 || (reinterpret_cast<uintptr_t>(output) & 3u)) return MIXQ_E_ALIGN;
    const void* zeros = mixq::zero_page();
    if (!zeros) return MIXQ_E_HIP;
    if (workspace && !aligned16(workspace)) workspace = nullptr, workspace_bytes = 0;
    return hip_rc(mixq::launch_w8a16_gemm(input, weight, scale, output, m, n, k, workspace, workspace_bytes, zeros, st));
}

int mixq_w8a16_gemm_forward(const void* input, const uint8_t* weight, const void* scale, void* output, int m, int n,
                            int k, void* stream)
{
    return mixq_w8a16_gemm_forward_ws(input, weight, scale, output, m, n, k, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------- enqueue ----
// decode batches that the weight-streaming skinny GEMM serves: the quantiser writes qA in that kernel's MFMA fragment order (1), else 0
static int enqueue_qa_frag(int64_t M, int64_t N, int64_t K, void* scratch, const void* W, const void* img)
{
    if (!(M <= 64 && K % 16 == 0 && N % 16 == 0 && mixq::qa_frag_enabled() && mixq::quant_frag_layout_supported((int)M, (int)K))) return 0;
    mixq::GemmParams probe{};
    probe.M = (int)M, probe.N = (int)N, probe.K = (int)K, probe.O = kNumOutliers;
    probe.splitk_ws = scratch;
    probe.B = static_cast<const int8_t*>(W);
    probe.b_image = img; // (a registered weight image widens the skinny kernel's range)
    probe.a_frag = 1;    // ("if the quantiser writes the fragment-major image, does the skinny kernel take the problem?")
    return mixq::gemm_takes_skinny(probe, mixq::EPI_DEQUANT) ? 1 : 0;
}

static int enqueue_impl(const mixq_handle* h, const mixq_tensor_desc* inputDesc, const void* const* inputs,
                        void* const* outputs, void* workspace, void* stream, void* ev_gemm_start, void* ev_gemm_stop)
{
    if (!h || !inputDesc || !inputs || !outputs) return MIXQ_E_BADARG;
    const mixq_tensor_desc& a = inputDesc[0];
    if (a.nbDims < 1 || a.nbDims > MIXQ_MAX_DIMS || inputDesc[1].nbDims < 1) return MIXQ_E_BADARG;
    int64_t M = 1;
    for (int i = 0; i < a.nbDims - 1; ++i) M *= a.d[i];      // TsinghuaMixQPlugin.cpp:390-394
    const int64_t K = a.d[a.nbDims - 1];                     // :396
    const int64_t N = inputDesc[1].d[0];                     // :399
    if (M < 0 || K <= 0 || N <= 0 || M > INT32_MAX || K > INT32_MAX || N > INT32_MAX) return MIXQ_E_BADARG;
    if (M == 0) return MIXQ_OK;
    for (int i = 0; i < 7; ++i)
        if (!inputs[i]) return MIXQ_E_BADARG;
    if (!outputs[0]) return MIXQ_E_BADARG;

    void* Out = outputs[0];
    const void* A = inputs[0];
    const int8_t* W = static_cast<const int8_t*>(inputs[1]);
    const void* scale_b = inputs[2];
    const void* fp_weight = inputs[3];
    const int32_t* ind = static_cast<const int32_t*>(inputs[4]);
    const uint8_t* q_weight = static_cast<const uint8_t*>(inputs[5]);
    const void* scaling_factors = inputs[6];

    if (M > kSmallMFastPath) {
        if (!workspace) return MIXQ_E_WORKSPACE;
        // workspace carve, same order and 128-B alignment as TsinghuaMixQPlugin.cpp:404-421
        uintptr_t base = align_up(reinterpret_cast<uintptr_t>(workspace), kWorkspaceAlign);
        int8_t* qA = reinterpret_cast<int8_t*>(base);
        base = align_up(base + qa_region_bytes(M, K), kWorkspaceAlign);
        void* sA = reinterpret_cast<void*>(base);
        base = align_up(base + (size_t)M * sizeof(uint16_t), kWorkspaceAlign);
        void* fpA = reinterpret_cast<void*>(base);

        // Mid-size and small problems split K over several workgroups per tile, which hand their partial sums over
        // through scratch behind fpA (gemm_pp_kernels.hip, gemm_kernels.hip).  mixq_workspace_size reserves the 256x256
        // form's scratch from 129 rows on where its plan applies (the mid-M deep form's where its table does), the small-tile form's otherwise.  The hand-over words at its start are
        // cleared on every call -- the workspace is shared with whatever else the engine runs -- by the quantiser, which
        // runs one launch earlier anyway.
        hipStream_t st = static_cast<hipStream_t>(stream);
        void* scratch = nullptr;
        const size_t scratch_bytes = enqueue_scratch_bytes(M, N, K);
        if (scratch_bytes) {
            base = align_up(base + (size_t)M * (size_t)kNumOutliers * sizeof(uint16_t), kWorkspaceAlign);
            scratch = reinterpret_cast<void*>(base);
        }
        if (K % 8) return MIXQ_E_SHAPE;
        if (!aligned16(A)) return MIXQ_E_ALIGN;
        // decode batches that the weight-streaming skinny GEMM serves: the quantiser writes qA in that kernel's MFMA fragment
        // order, so that each of its qA loads is one contiguous 1-KiB read (profiles/r03_small_m_timeline.txt)
        // (Large calls keep the row-major image: a K-slice-major one -- 8 consecutive rows of a 128-byte slice contiguous, one LDS-DMA
        //  instruction = one 1-KiB read -- was built and measured on one box: GEMM -0.3 %, quantiser +7.6 % (its row becomes 32
        //  scattered 128-byte stores), prefill tokens/s -0.35 %: docs/LAB_NOTEBOOK.md R3.10.)
        const void* img = M <= 64 ? mixq::resolve_weight_image(W, (int)N, (int)K, st) : nullptr; // (ONE lookup per call: probe and launch agree)
        const int frag = enqueue_qa_frag(M, N, K, scratch, W, img);
        int rc = hip_rc(mixq::launch_quant_extract(const_cast<void*>(A), qA, sA, fpA, ind, (int)M, (int)K, kNumOutliers,
                                                   false, st, scratch, frag));
        if (rc != MIXQ_OK) return rc;
        if (ev_gemm_start && hipEventRecord(static_cast<hipEvent_t>(ev_gemm_start), st) != hipSuccess) return MIXQ_E_HIP;
        rc = gemm_mixed_impl(qA, W, sA, scale_b, fpA, fp_weight, Out, (int)M, (int)N, (int)K, kNumOutliers, scratch,
                             scratch_bytes, stream, frag, img, true);
        if (ev_gemm_stop && hipEventRecord(static_cast<hipEvent_t>(ev_gemm_stop), st) != hipSuccess) return MIXQ_E_HIP;
        return rc;
    }
    return mixq_w8a16_gemm_forward(A, q_weight, scaling_factors, Out, (int)M, (int)N, (int)K, stream);
}

// What mixq_enqueue would launch for a call with M rows on an [N, K] layer, as text (host only, no GPU needed; include/mixq.h).
int mixq_describe_plan(int64_t M, int64_t N, int64_t K, int have_weight_image, char* buf, size_t len)
{
    if (!buf || len == 0 || M <= 0 || N <= 0 || K <= 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return MIXQ_E_BADARG;
    if (M <= kSmallMFastPath) {
        snprintf(buf, len, "w8a16: one launch on qweight (fpA_intB GEMV / skinny form)");
        return MIXQ_OK;
    }
    static int dummy_scratch, dummy_image;
    void* scratch = enqueue_scratch_bytes(M, N, K) ? &dummy_scratch : nullptr;
    const void* img = have_weight_image && M <= 64 && K % 64 == 0 && N % 16 == 0 ? &dummy_image : nullptr;
    mixq::GemmParams p{};
    p.M = (int)M, p.N = (int)N, p.K = (int)K, p.O = kNumOutliers;
    p.splitk_ws = scratch;
    p.b_image = img;
    p.a_frag = enqueue_qa_frag(M, N, K, scratch, &dummy_image, img);
    char plan[160];
    mixq::describe_gemm_plan(p, mixq::EPI_DEQUANT, plan, sizeof plan);
    snprintf(buf, len, "quantise + extract (1 launch), then %s", plan);
    return MIXQ_OK;
}

int mixq_enqueue(const mixq_handle* h, const mixq_tensor_desc* inputDesc, const mixq_tensor_desc* /*outputDesc*/,
                 const void* const* inputs, void* const* outputs, void* workspace, void* stream)
{
    return enqueue_impl(h, inputDesc, inputs, outputs, workspace, stream, nullptr, nullptr);
}

int mixq_enqueue_profiled(const mixq_handle* h, const mixq_tensor_desc* inputDesc,
                          const mixq_tensor_desc* /*outputDesc*/, const void* const* inputs, void* const* outputs,
                          void* workspace, void* stream, void* ev_gemm_start, void* ev_gemm_stop)
{
    return enqueue_impl(h, inputDesc, inputs, outputs, workspace, stream, ev_gemm_start, ev_gemm_stop);
}

// ------------------------------------------------------------------------------ multi-GPU ----
int mixq_tp_fused_supported(int64_t M, int64_t N_local, int64_t K)
{
    if (M <= 0 || N_local <= 0 || K <= 0 || M > INT32_MAX || N_local > INT32_MAX || K > INT32_MAX) return 0;
    return mixq::gemm_tp_fused_supported((int)M, (int)N_local, (int)K, kNumOutliers) ? 1 : 0;
}

int mixq_tp_flag_words(int64_t M) { return M > 0 && M <= INT32_MAX ? mixq::tp_flag_words((int)M) : 1; }

int mixq_enqueue_tp(const mixq_handle* h, const mixq_tensor_desc* inputDesc, const void* const* inputs, void* workspace,
                    const mixq_tp_epilogue* tp, void* stream)
{
    if (!h || !inputDesc || !inputs || !tp) return MIXQ_E_BADARG;
    const mixq_tensor_desc& a = inputDesc[0];
    if (a.nbDims < 1 || a.nbDims > MIXQ_MAX_DIMS || inputDesc[1].nbDims < 1) return MIXQ_E_BADARG;
    int64_t M = 1;
    for (int i = 0; i < a.nbDims - 1; ++i) M *= a.d[i];
    const int64_t K = a.d[a.nbDims - 1], N = inputDesc[1].d[0];
    if (M <= 0 || K <= 0 || N <= 0 || M > INT32_MAX || K > INT32_MAX || N > INT32_MAX) return MIXQ_E_BADARG;
    if (tp->ndst < 1 || tp->ndst > mixq::kTpMaxPeers || tp->n_total <= 0 || tp->col0 < 0 || tp->col0 + N > tp->n_total ||
        !tp->counters)
        return MIXQ_E_BADARG;
    if (tp->n_total % 8 || tp->col0 % 8 || N % 16 || K % 16) return MIXQ_E_SHAPE;
    if (!mixq::gemm_tp_fused_supported((int)M, (int)N, (int)K, kNumOutliers)) return MIXQ_E_SHAPE; // caller: enqueue + push
    for (int i = 0; i < 5; ++i)
        if (!inputs[i]) return MIXQ_E_BADARG;
    if (!workspace) return MIXQ_E_WORKSPACE;
    if (!aligned16(inputs[0]) || !aligned16(inputs[1]) || !aligned16(inputs[2]) || !aligned16(inputs[3])) return MIXQ_E_ALIGN;
    uintptr_t base = align_up(reinterpret_cast<uintptr_t>(workspace), kWorkspaceAlign);
    int8_t* qA = reinterpret_cast<int8_t*>(base);
    base = align_up(base + (size_t)M * (size_t)K, kWorkspaceAlign);
    void* sA = reinterpret_cast<void*>(base);
    base = align_up(base + (size_t)M * sizeof(uint16_t), kWorkspaceAlign);
    void* fpA = reinterpret_cast<void*>(base);
    hipStream_t st = static_cast<hipStream_t>(stream);
    mixq::GemmParams p{};
    p.A = qA, p.B = static_cast<const int8_t*>(inputs[1]);
    p.sA = static_cast<const uint16_t*>(sA), p.sW = static_cast<const uint16_t*>(inputs[2]);
    p.fpA = static_cast<const uint16_t*>(fpA), p.fpW = static_cast<const uint16_t*>(inputs[3]);
    p.O = kNumOutliers, p.M = (int)M, p.N = (int)N, p.K = (int)K;
    p.zeros = mixq::zero_page();
    if (!p.zeros) return MIXQ_E_HIP;
    p.tp.ndst = tp->ndst, p.tp.ldd = tp->n_total, p.tp.seq = tp->seq;
    p.tp.counters = static_cast<unsigned*>(tp->counters);
    for (int r = 0; r < tp->ndst; ++r) {
        if (!tp->dst_bases[r] || !tp->dst_flags[r] || !aligned16(tp->dst_bases[r])) return MIXQ_E_BADARG;
        p.tp.base[r] = static_cast<char*>(tp->dst_bases[r]) + (size_t)tp->col0 * sizeof(uint16_t);
        p.tp.flag[r] = static_cast<unsigned*>(tp->dst_flags[r]);
    }
    int rc = hip_rc(mixq::launch_quant_extract(const_cast<void*>(inputs[0]), qA, sA, fpA, static_cast<const int32_t*>(inputs[4]),
                                               (int)M, (int)K, kNumOutliers, false, st, nullptr));
    if (rc != MIXQ_OK) return rc;
    return hip_rc(mixq::launch_gemm_pp_tp(p, st));
}


int mixq_tp_buffer_alloc(size_t bytes, int mem_kind, void** dev_ptr, void* ipc_handle_64)
{
    if (!dev_ptr || !ipc_handle_64 || bytes == 0) return MIXQ_E_BADARG;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size is part of the ABI");
    void* p = nullptr;
    // Memory a REMOTE GPU writes while the local GPU reads / polls it must be fine-grained (or uncached): system-scope
    // atomics are only specified there; plain hipMalloc memory is coarse-grained (coherent at kernel boundaries only).
    // No silent downgrade: the caller asked for a kind and gets it or an error.
    hipError_t e;
    switch (mem_kind) {
    case MIXQ_TP_MEM_COARSE: e = hipMalloc(&p, bytes); break;
    case MIXQ_TP_MEM_FINEGRAINED: e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained); break;
    case MIXQ_TP_MEM_UNCACHED: e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached); break;
    default: return MIXQ_E_BADARG;
    }
    if (e != hipSuccess) return MIXQ_E_HIP;
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        return MIXQ_E_HIP;
    }
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
        (void)hipFree(p);
        return MIXQ_E_HIP;
    }
    std::memcpy(ipc_handle_64, &h, sizeof(h));
    *dev_ptr = p;
    return MIXQ_OK;
}

int mixq_tp_buffer_open(const void* ipc_handle_64, void** dev_ptr)
{
    if (!ipc_handle_64 || !dev_ptr) return MIXQ_E_BADARG;
    hipIpcMemHandle_t h;
    std::memcpy(&h, ipc_handle_64, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return MIXQ_E_HIP;
    *dev_ptr = p;
    return MIXQ_OK;
}

int mixq_tp_buffer_close(void* dev_ptr) { return dev_ptr ? hip_rc(hipIpcCloseMemHandle(dev_ptr)) : MIXQ_E_BADARG; }
int mixq_tp_buffer_free(void* dev_ptr) { return dev_ptr ? hip_rc(hipFree(dev_ptr)) : MIXQ_E_BADARG; }

int mixq_tp_status_alloc(void** host_ptr, void** dev_ptr)
{
    if (!host_ptr || !dev_ptr) return MIXQ_E_BADARG;
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return MIXQ_E_HIP;
    std::memset(h, 0, 64);
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
        (void)hipHostFree(h);
        return MIXQ_E_HIP;
    }
    *host_ptr = h, *dev_ptr = d;
    return MIXQ_OK;
}

int mixq_tp_status_free(void* host_ptr) { return host_ptr ? hip_rc(hipHostFree(host_ptr)) : MIXQ_E_BADARG; }

int mixq_tp_push_columns(const void* src, void* const* dst_bases, void* const* dst_flags, int ndst, int M, int n_local,
                         int N, int col0, uint32_t seq, int nflags, void* done_counter, void* stream)
{
    if (!dst_bases || !dst_flags || !done_counter || ndst < 1 || ndst > 8 || M < 0 || n_local <= 0 || N <= 0 ||
        col0 < 0 || col0 + n_local > N || nflags < 1 || nflags > mixq::kTpFlagWords || (M > 0 && !src))
        return MIXQ_E_BADARG;
    if (n_local % 8 || N % 8 || col0 % 8) return MIXQ_E_SHAPE;
    if (M > 0 && !aligned16(src)) return MIXQ_E_ALIGN;
    unsigned* flags[8];
    for (int r = 0; r < ndst; ++r) {
        if (!dst_bases[r] || !dst_flags[r] || !aligned16(dst_bases[r])) return MIXQ_E_BADARG;
        flags[r] = static_cast<unsigned*>(dst_flags[r]);
    }
    return hip_rc(mixq::launch_tp_push(src, dst_bases, flags, ndst, M, n_local, N, col0, seq, nflags,
                                       static_cast<unsigned*>(done_counter), static_cast<hipStream_t>(stream)));
}

int mixq_tp_wait(const void* flags, int nprod, int word0, int nwords, uint32_t seq, void* status_dev, int trap_on_timeout,
                 uint32_t patience_ms, void* stream)
{
    if (!flags || !status_dev || nprod < 1 || nprod > 8 || word0 < 0 || nwords < 1 || word0 + nwords > mixq::kTpFlagWords)
        return MIXQ_E_BADARG;
    return hip_rc(mixq::launch_tp_wait(static_cast<const unsigned*>(flags), nprod, word0, nwords, seq,
                                       static_cast<unsigned*>(status_dev), trap_on_timeout, patience_ms,
                                       static_cast<hipStream_t>(stream)));
}

// Capturable form of the same gather (tp_kernels.hip): the call number lives in `seq_word` (one zeroed device word of this rank,
// bumped by mixq_tp_wait_seq), every rank has ONE destination buffer, and its reuse is acknowledged explicitly.
int mixq_tp_arrive(void* const* peer_ack_words, const void* own_ack_words, int npeer, const void* seq_word, void* status_dev,
                   int trap_on_timeout, uint32_t patience_ms, void* stream)
{
    if (!peer_ack_words || !own_ack_words || !seq_word || !status_dev || npeer < 1 || npeer > 8) return MIXQ_E_BADARG;
    unsigned* acks[8];
    for (int r = 0; r < npeer; ++r) {
        if (!peer_ack_words[r]) return MIXQ_E_BADARG;
        acks[r] = static_cast<unsigned*>(peer_ack_words[r]);
    }
    return hip_rc(mixq::launch_tp_arrive(acks, static_cast<const unsigned*>(own_ack_words), npeer,
                                         static_cast<const unsigned*>(seq_word), static_cast<unsigned*>(status_dev),
                                         trap_on_timeout, patience_ms, static_cast<hipStream_t>(stream)));
}

int mixq_tp_push_columns_seq(const void* src, void* const* dst_bases, void* const* dst_flags, int ndst, int M, int n_local,
                             int N, int col0, const void* seq_word, void* done_counter, const void* status_dev, void* stream)
{
    if (!dst_bases || !dst_flags || !done_counter || !seq_word || ndst < 1 || ndst > 8 || M < 0 || n_local <= 0 || N <= 0 ||
        col0 < 0 || col0 + n_local > N || (M > 0 && !src))
        return MIXQ_E_BADARG;
    if (n_local % 8 || N % 8 || col0 % 8) return MIXQ_E_SHAPE;
    if (M > 0 && !aligned16(src)) return MIXQ_E_ALIGN;
    unsigned* flags[8];
    for (int r = 0; r < ndst; ++r) {
        if (!dst_bases[r] || !dst_flags[r] || !aligned16(dst_bases[r])) return MIXQ_E_BADARG;
        flags[r] = static_cast<unsigned*>(dst_flags[r]);
    }
    return hip_rc(mixq::launch_tp_push(src, dst_bases, flags, ndst, M, n_local, N, col0, 0u, 1,
                                       static_cast<unsigned*>(done_counter), static_cast<hipStream_t>(stream),
                                       static_cast<const unsigned*>(seq_word), static_cast<const unsigned*>(status_dev)));
}

int mixq_tp_wait_seq(const void* flags, int nprod, void* seq_word, void* status_dev, int trap_on_timeout, uint32_t patience_ms,
                     void* stream)
{
    if (!flags || !status_dev || !seq_word || nprod < 1 || nprod > 8) return MIXQ_E_BADARG;
    return hip_rc(mixq::launch_tp_wait(static_cast<const unsigned*>(flags), nprod, 0, 1, 0u, static_cast<unsigned*>(status_dev),
                                       trap_on_timeout, patience_ms, static_cast<hipStream_t>(stream),
                                       static_cast<unsigned*>(seq_word)));
}

// ------------------------------------------------------------------------------------- host helpers ----
static inline size_t swap12(size_t x) { return (x & ~(size_t)3) | ((x & 1) << 1) | ((x & 2) >> 1); }

// physical byte offset of logical element (k, n) in the interleaved image (see decode_kernels.hip header)
static inline size_t eetq_offset(size_t k, size_t n, size_t K)
{
    static const int inv[16] = {0, 1, 4, 5, 8, 9, 12, 13, 2, 3, 6, 7, 10, 11, 14, 15}; // inverse of perm16
    const size_t tb = k / 64, g = (k % 64) / 16, t = inv[k % 16];
    const size_t xprime = g * 16 + t;
    return (n / 2) * 2 * K + tb * 128 + (n % 2) * 64 + swap12(xprime);
}

int mixq_preprocess_weights_int8(uint8_t* preprocessed, const int8_t* row_major, size_t rows, size_t cols)
{
    if (!preprocessed || !row_major) return MIXQ_E_BADARG;
    if (rows % 64 || cols % 2) return MIXQ_E_SHAPE;
    for (size_t k = 0; k < rows; ++k)
        for (size_t n = 0; n < cols; ++n)
            preprocessed[eetq_offset(k, n, rows)] = (uint8_t)((int)row_major[k * cols + n] + 128);
    return MIXQ_OK;
}

int mixq_unprocess_weights_int8(int8_t* row_major, const uint8_t* preprocessed, size_t rows, size_t cols)
{
    if (!preprocessed || !row_major) return MIXQ_E_BADARG;
    if (rows % 64 || cols % 2) return MIXQ_E_SHAPE;
    for (size_t k = 0; k < rows; ++k)
        for (size_t n = 0; n < cols; ++n)
            row_major[k * cols + n] = (int8_t)((int)preprocessed[eetq_offset(k, n, rows)] - 128);
    return MIXQ_OK;
}

} // extern "C"
