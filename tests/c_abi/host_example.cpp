// The same host as host_example.c, written against include/mixq_plugin.hpp: the reference's two classes
// (TsinghuaMixQPlugin.h:34-115) by their own method names.  Test infrastructure (tests/test_gpu_c_host.py).
//   usage: host_example_cpp <dir> <M> <N> <K>      exit code 0 = within 1e-3 of the expected output (max-normalised)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

#include "mixq_plugin.hpp"

using namespace mixq_plugin;

static float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int exp = (h >> 10) & 0x1f;
    const uint32_t man = h & 0x3ffu;
    if (exp == 0) return std::ldexp((float)man, -24) * (sign ? -1.f : 1.f);
    uint32_t bits = exp == 31 ? (sign | 0x7f800000u | (man << 13)) : (sign | ((uint32_t)(exp + 112) << 23) | (man << 13));
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

static std::vector<char> read_file(const std::string& path, size_t bytes)
{
    std::ifstream f(path, std::ios::binary);
    std::vector<char> v(bytes);
    if (!f || !f.read(v.data(), (std::streamsize)bytes)) {
        std::fprintf(stderr, "cannot read %zu bytes of %s\n", bytes, path.c_str());
        std::exit(2);
    }
    return v;
}

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); std::exit(3); } } while (0)

static void* to_device(const std::vector<char>& host)
{
    void* d = nullptr;
    HIP_OK(hipMalloc(&d, host.size()));
    HIP_OK(hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
    return d;
}

static PluginTensorDesc desc(int64_t d0, int64_t d1 = 0)
{
    PluginTensorDesc t;
    std::memset(&t, 0, sizeof t);
    t.nbDims = d1 ? 2 : 1, t.d[0] = d0, t.d[1] = d1, t.type = MIXQ_TYPE_HALF, t.format = MIXQ_FORMAT_LINEAR, t.scale = 1.f;
    return t;
}

int main(int argc, char** argv)
{
    if (argc != 5) return 2;
    const std::string dir = argv[1];
    const int32_t M = std::atoi(argv[2]), N = std::atoi(argv[3]), K = std::atoi(argv[4]);
    const size_t mk = (size_t)M * K, nk = (size_t)N * K, mn = (size_t)M * N;
    if (mixq_abi_version() != MIXQ_ABI_VERSION) return 9;   // a library built from another revision of mixq.h
    if (!initOpenAiTritonPlugins(nullptr, "tensorrt_llm")) return 4;

    void* A = to_device(read_file(dir + "/A.f16", mk * 2));
    void* weight = to_device(read_file(dir + "/weight.i8", nk));
    void* sW = to_device(read_file(dir + "/weights_scaling_factor.f16", (size_t)N * 2));
    void* fpW = to_device(read_file(dir + "/fp_weight.f16", (size_t)N * 256));
    void* ind = to_device(read_file(dir + "/fp_ind.i32", 512));
    void* qweight = to_device(read_file(dir + "/qweight.u8", nk));
    const std::vector<char> want_raw = read_file(dir + "/want.f16", mn * 2);
    const uint16_t* want = reinterpret_cast<const uint16_t*>(want_raw.data());
    void* Out = nullptr;
    HIP_OK(hipMalloc(&Out, mn * 2));

    MixQPluginCreator creator;
    creator.setPluginNamespace("tensorrt_llm");
    const PluginField fields[3] = {{"m", &M, MIXQ_FIELD_INT32, 1}, {"n", &N, MIXQ_FIELD_INT32, 1}, {"k", &K, MIXQ_FIELD_INT32, 1}};
    const PluginFieldCollection fc = {3, fields};
    { // getFieldNames (TsinghuaMixQPlugin.cpp:890-893): the advertised table is the reference's "mm", "mn", "mk" (INT32, no data, length -1)
        const PluginFieldCollection* adv = creator.getFieldNames();
        if (!adv || adv->nbFields != 3 || std::string(adv->fields[0].name) != "mm" || std::string(adv->fields[1].name) != "mn" ||
            std::string(adv->fields[2].name) != "mk" || adv->fields[0].type != MIXQ_FIELD_INT32 || adv->fields[0].data != nullptr ||
            adv->fields[2].length != -1)
            return 5;
    }
    MixQPlugin* plugin = creator.createPlugin("layer", &fc);
    if (!plugin || plugin->initialize() != 0 || std::string(plugin->getPluginType()) != "MixQ" || plugin->getNbOutputs() != 1) return 5;
    const PluginFieldCollection bad = {2, fields};
    if (creator.createPlugin("layer", &bad) != nullptr) return 5; // a collection without "k" is refused

    PluginTensorDesc in[8] = {desc(M, K), desc(N, K / 2), desc(N), desc(N, 128), desc(256), desc(K, N / 2), desc(N), desc(M, N)};
    for (int pos = 0; pos < 8; ++pos)
        if (!plugin->supportsFormatCombination(pos, in, 7, 1)) return 6;
    PluginTensorDesc out;
    if (plugin->getOutputDimensions(0, in, 7, &out) != 0 || out.nbDims != 2 || out.d[0] != M || out.d[1] != N) return 6;
    plugin->configurePlugin(in, 7, &out, 1);
    const size_t ws_bytes = plugin->getWorkspaceSize(in, 7, &out, 1);
    void* ws = nullptr;
    HIP_OK(hipMalloc(&ws, ws_bytes ? ws_bytes : 16));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    const void* inputs[7] = {A, weight, sW, fpW, ind, qweight, sW};
    void* outputs[1] = {Out};
    if (plugin->enqueue(in, &out, inputs, outputs, ws, st) != 0) return 7;
    HIP_OK(hipStreamSynchronize(st));
    std::vector<uint16_t> got(mn), again(mn);
    HIP_OK(hipMemcpy(got.data(), Out, mn * 2, hipMemcpyDeviceToHost));
    double max_want = 0., max_diff = 0.;
    for (size_t i = 0; i < mn; ++i) {
        const double w = half_to_float(want[i]), g = half_to_float(got[i]);
        max_want = std::fabs(w) > max_want ? std::fabs(w) : max_want;
        max_diff = std::fabs(g - w) > max_diff ? std::fabs(g - w) : max_diff;
    }
    const double rel = max_diff / (max_want > 0. ? max_want : 1.);

    // clone() and serialize() -> deserializePlugin(): same bits from both
    std::vector<char> blob(plugin->getSerializationSize());
    plugin->serialize(blob.data());
    MixQPlugin* twins[2] = {plugin->clone(), creator.deserializePlugin("layer", blob.data(), blob.size())};
    bool same = blob.size() == 12;
    for (MixQPlugin* t : twins) {
        if (!t || t->initialize() != 0) return 8;
        HIP_OK(hipMemset(Out, 0, mn * 2));
        if (t->enqueue(in, &out, inputs, outputs, ws, st) != 0) return 8;
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipMemcpy(again.data(), Out, mn * 2, hipMemcpyDeviceToHost));
        same = same && std::memcmp(got.data(), again.data(), mn * 2) == 0 && std::string(t->getPluginNamespace()) == "tensorrt_llm";
        t->terminate();
        t->destroy();
    }
    std::printf("C++ host: %s, M=%d N=%d K=%d, workspace %zu bytes, rel err %.3g, clone and deserialised twin bit-identical: %d\n",
                mixq_version(), (int)M, (int)N, (int)K, ws_bytes, rel, (int)same);
    plugin->terminate();
    plugin->destroy();
    return (rel < 1e-3 && same) ? 0 : 1;
}
