/* A plain C99 host of the C ABI (include/mixq.h): the plugin lifecycle the reference's C++ host runs through TensorRT
 * (TsinghuaMixQPlugin.h:34-89: creator -> initialize -> getWorkspaceSize -> enqueue -> serialize / deserialize -> destroy),
 * on raw device pointers and a HIP stream -- no Python, no torch, no C++.  Test infrastructure: tests/test_gpu_c_host.py
 * writes the seven input tensors of one MixQ linear and the oracle's output as raw files, compiles this file with gcc and
 * runs it.
 *   usage: host_example <dir> <M> <N> <K>      exit code 0 = within 1e-3 of the expected output (max-normalised) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

#include "mixq.h"

static float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal */
            exp = 127 - 15 + 1;
            while (!(man & 0x400u)) man <<= 1, --exp;
            bits = sign | (exp << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static void* read_file(const char* dir, const char* name, size_t bytes)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    void* p = malloc(bytes);
    if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "%s: short read (want %zu bytes)\n", path, bytes); exit(2); }
    fclose(f);
    return p;
}

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(3); } } while (0)

static void* to_device(const void* host, size_t bytes)
{
    void* d = NULL;
    HIP_OK(hipMalloc(&d, bytes));
    HIP_OK(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    return d;
}

static mixq_tensor_desc desc2(int64_t d0, int64_t d1)
{
    mixq_tensor_desc t;
    memset(&t, 0, sizeof t);
    t.nbDims = d1 ? 2 : 1, t.d[0] = d0, t.d[1] = d1, t.type = MIXQ_TYPE_HALF, t.format = MIXQ_FORMAT_LINEAR, t.scale = 1.f;
    return t;
}

int main(int argc, char** argv)
{
    if (argc != 5) { fprintf(stderr, "usage: %s <dir> <M> <N> <K>\n", argv[0]); return 2; }
    const char* dir = argv[1];
    const int32_t M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]);
    const size_t mk = (size_t)M * K, nk = (size_t)N * K, mn = (size_t)M * N;

    if (mixq_abi_version() != MIXQ_ABI_VERSION) return 9; /* a library built from another revision of mixq.h */
    /* the reference's loaders call this first (plugin.py:34-43) */
    if (!initOpenAiTritonPlugins(NULL, "tensorrt_llm") || !mixq_registry_has_creator("MixQ", "1", "tensorrt_llm")) return 4;

    void* A = to_device(read_file(dir, "A.f16", mk * 2), mk * 2);
    void* weight = to_device(read_file(dir, "weight.i8", nk), nk);
    void* sW = to_device(read_file(dir, "weights_scaling_factor.f16", (size_t)N * 2), (size_t)N * 2);
    void* fpW = to_device(read_file(dir, "fp_weight.f16", (size_t)N * 128 * 2), (size_t)N * 128 * 2);
    void* ind = to_device(read_file(dir, "fp_ind.i32", 512), 512);
    void* qweight = to_device(read_file(dir, "qweight.u8", nk), nk);
    uint16_t* want = (uint16_t*)read_file(dir, "want.f16", mn * 2);
    void* Out = NULL;
    HIP_OK(hipMalloc(&Out, mn * 2));

    /* MixQPluginCreator::createPlugin (TsinghuaMixQPlugin.cpp:895-933): fields "m", "n", "k" */
    const mixq_plugin_field fields[3] = {{"m", &M, MIXQ_FIELD_INT32, 1}, {"n", &N, MIXQ_FIELD_INT32, 1}, {"k", &K, MIXQ_FIELD_INT32, 1}};
    mixq_handle* h = mixq_create_from_fields(fields, 3);
    if (!h || mixq_initialize(h) != 0) return 5;
    const size_t ws_bytes = mixq_workspace_size(h, M, N, K);
    void* ws = NULL;
    HIP_OK(hipMalloc(&ws, ws_bytes ? ws_bytes : 16));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    mixq_tensor_desc in[7], out = desc2(M, N);
    in[0] = desc2(M, K), in[1] = desc2(N, K / 2), in[2] = desc2(N, 0), in[3] = desc2(N, 128), in[4] = desc2(256, 0);
    in[5] = desc2(K, N / 2), in[6] = desc2(N, 0);
    const void* inputs[7] = {A, weight, sW, fpW, ind, qweight, sW};
    void* outputs[1] = {Out};
    int rc = mixq_enqueue(h, in, &out, inputs, outputs, ws, (void*)st);
    if (rc != 0) { fprintf(stderr, "mixq_enqueue: %s\n", mixq_error_string(rc)); return 6; }
    HIP_OK(hipStreamSynchronize(st));
    uint16_t* got = (uint16_t*)malloc(mn * 2);
    HIP_OK(hipMemcpy(got, Out, mn * 2, hipMemcpyDeviceToHost));

    double max_want = 0., max_diff = 0.;
    for (size_t i = 0; i < mn; ++i) {
        const double w = half_to_float(want[i]), g = half_to_float(got[i]);
        if (fabs(w) > max_want) max_want = fabs(w);
        if (fabs(g - w) > max_diff) max_diff = fabs(g - w);
    }
    const double rel = max_diff / (max_want > 0. ? max_want : 1.);

    /* engine serialisation carries 3 x int32 (TsinghuaMixQPlugin.cpp:227-234, 813-820): a deserialised clone gives the same bits */
    char blob[64];
    const size_t blob_bytes = mixq_serialization_size(h);
    if (blob_bytes != 12) return 7;
    mixq_serialize(h, blob);
    mixq_handle* h2 = mixq_deserialize(blob, blob_bytes);
    if (!h2 || mixq_initialize(h2) != 0) return 7;
    HIP_OK(hipMemset(Out, 0, mn * 2));
    rc = mixq_enqueue(h2, in, &out, inputs, outputs, ws, (void*)st);
    HIP_OK(hipStreamSynchronize(st));
    uint16_t* again = (uint16_t*)malloc(mn * 2);
    HIP_OK(hipMemcpy(again, Out, mn * 2, hipMemcpyDeviceToHost));
    const int same = rc == 0 && memcmp(got, again, mn * 2) == 0;

    printf("C host: %s, M=%d N=%d K=%d, workspace %zu bytes, rel err %.3g, deserialised clone bit-identical: %d\n", mixq_version(),
           (int)M, (int)N, (int)K, ws_bytes, rel, same);
    mixq_terminate(h2);
    mixq_destroy(h2);
    mixq_terminate(h);
    mixq_destroy(h);
    return (rel < 1e-3 && same) ? 0 : 1;
}
