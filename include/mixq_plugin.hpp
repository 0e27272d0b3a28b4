/* mixq_plugin.hpp -- C++11 mirror of the reference's plugin classes over the C ABI of mixq.h (header-only).
 *
 * For a C++ runtime that used to hold `nvinfer1::IPluginV2DynamicExt*` objects of the reference
 * (TsinghuaMixQPlugin.h:34-115, TsinghuaMixQPlugin.cpp:217-951): the same two classes with the same method names and the
 * same meaning, minus the TensorRT base classes (there is no TensorRT on MI355X).  Differences are the ones of the C ABI:
 * descriptors are the POD `mixq_tensor_desc` (a mirror of nvinfer1::PluginTensorDesc), the stream is a hipStream_t passed
 * as void*, getOutputDimensions takes concrete dimensions instead of IExprBuilder expressions, and enqueue reports errors
 * (the reference always returns 0).  Nothing here touches the device; every call forwards to the library.
 *
 *   mixq_plugin::MixQPluginCreator creator;                        // registered name / version: "MixQ" / "1"
 *   mixq_plugin::MixQPlugin* p = creator.createPlugin("layer0.qkv", &fc);   // fields "m", "n", "k" (INT32)
 *   p->initialize();
 *   size_t ws = p->getWorkspaceSize(in_max, 7, out_max, 1);
 *   p->enqueue(in, out, inputs, outputs, workspace, stream);
 *   p->destroy();
 */
#ifndef MIXQ_PLUGIN_HPP_
#define MIXQ_PLUGIN_HPP_

#include <cstddef>
#include <cstdint>
#include <new>
#include <string>

#include "mixq.h"

namespace mixq_plugin {

typedef mixq_tensor_desc PluginTensorDesc;  /* nvinfer1::PluginTensorDesc */
typedef mixq_plugin_field PluginField;      /* nvinfer1::PluginField */
struct PluginFieldCollection {              /* nvinfer1::PluginFieldCollection */
    int32_t nbFields;
    const PluginField* fields;
};

class MixQPlugin {
public:
    MixQPlugin(int m, int n, int k) : h_(mixq_create(m, n, k)) {}                        /* .cpp:217-225 */
    MixQPlugin(const void* data, std::size_t length) : h_(mixq_deserialize(data, length)) {} /* .cpp:227-234 */
    bool valid() const noexcept { return h_ != nullptr; }

    /* IPluginV2DynamicExt */
    MixQPlugin* clone() const noexcept                                                   /* .cpp:237-242 */
    {
        mixq_handle* c = mixq_clone(h_);
        MixQPlugin* p = c ? new (std::nothrow) MixQPlugin(c) : nullptr;
        if (p) p->workspace_max_ = workspace_max_;   /* the reference's clone copies m_workspaceMaxSize (.cpp:236-241) */
        else if (c) mixq_destroy(c);
        return p;
    }
    /* out = inputs[0].dims with the last dimension replaced by inputs[1].d[0] (.cpp:244-261); 0 = ok */
    int getOutputDimensions(int outputIndex, const PluginTensorDesc* inputs, int nbInputs, PluginTensorDesc* out) const noexcept
    {
        return mixq_get_output_dimensions(h_, outputIndex, inputs, nbInputs, out);
    }
    bool supportsFormatCombination(int pos, const PluginTensorDesc* inOut, int nbInputs, int nbOutputs) const noexcept /* .cpp:263-320 */
    {
        return mixq_supports_format_combination(h_, pos, inOut, nbInputs, nbOutputs) == 1;
    }
    /* configurePlugin (.cpp:325-349) only records the maximum shapes for getWorkspaceSize: kept here the same way */
    void configurePlugin(const PluginTensorDesc* inMax, int nbInputs, const PluginTensorDesc* /*outMax*/, int /*nbOutputs*/) noexcept
    {
        if (nbInputs >= 2) workspace_max_ = workspace_for(inMax);
    }
    std::size_t getWorkspaceSize(const PluginTensorDesc* inputs, int nbInputs, const PluginTensorDesc* /*outputs*/,
                                 int /*nbOutputs*/) const noexcept                     /* .cpp:351-378 */
    {
        const std::size_t now = nbInputs >= 2 ? workspace_for(inputs) : 0;
        return now > workspace_max_ ? now : workspace_max_;
    }
    int enqueue(const PluginTensorDesc* inputDesc, const PluginTensorDesc* outputDesc, const void* const* inputs,
                void* const* outputs, void* workspace, void* stream) const noexcept     /* TsinghuaMixQPlugin.h:53-54 */
    {
        return mixq_enqueue(h_, inputDesc, outputDesc, inputs, outputs, workspace, stream);
    }
    /* IPluginV2Ext */
    int getOutputDataType(int index, const int* /*inputTypes*/, int /*nbInputs*/) const noexcept /* .cpp:768-773 */
    {
        return mixq_get_output_data_type(h_, index);
    }
    /* IPluginV2 */
    const char* getPluginType() const noexcept { return mixq_plugin_type(); }            /* "MixQ" */
    const char* getPluginVersion() const noexcept { return mixq_plugin_version(); }      /* "1" */
    int getNbOutputs() const noexcept { return mixq_get_nb_outputs(h_); }
    int initialize() noexcept { return mixq_initialize(h_); }
    void terminate() noexcept { mixq_terminate(h_); }
    std::size_t getSerializationSize() const noexcept { return mixq_serialization_size(h_); } /* 12 */
    void serialize(void* buffer) const noexcept { mixq_serialize(h_, buffer); }
    void destroy() noexcept { delete this; }                                             /* .cpp:851-855 */
    void setPluginNamespace(const char* ns) noexcept { mixq_set_namespace(h_, ns); }
    const char* getPluginNamespace() const noexcept { return mixq_get_namespace(h_); }

    const mixq_handle* handle() const noexcept { return h_; } /* for the entries that have no method here (mixq_enqueue_tp ...) */

private:
    explicit MixQPlugin(mixq_handle* h) : h_(h) {}
    ~MixQPlugin() { if (h_) mixq_destroy(h_); }
    MixQPlugin(const MixQPlugin&);            /* clone() is the copy */
    MixQPlugin& operator=(const MixQPlugin&);
    std::size_t workspace_for(const PluginTensorDesc* in) const noexcept
    {
        int64_t M = 1;
        for (int i = 0; i + 1 < in[0].nbDims; ++i) M *= in[0].d[i];                      /* .cpp:390-394 */
        const int64_t K = in[0].nbDims > 0 ? in[0].d[in[0].nbDims - 1] : 0, N = in[1].d[0];
        return mixq_workspace_size(h_, M, N, K);
    }
    mixq_handle* h_;
    std::size_t workspace_max_ = 0;
};

class MixQPluginCreator {                                                                /* .cpp:868-960 */
public:
    MixQPluginCreator() { mFC.fields = mixq_get_field_names(&mFC.nbFields); }           /* .cpp:868-878: "mm", "mn", "mk" */
    const PluginFieldCollection* getFieldNames() noexcept { return &mFC; }              /* .cpp:890-893 */
    const char* getPluginName() const noexcept { return mixq_plugin_type(); }
    const char* getPluginVersion() const noexcept { return mixq_plugin_version(); }
    /* reads the INT32 fields "m", "n", "k" (.cpp:906-919); nullptr on a malformed collection (the reference catches and logs) */
    MixQPlugin* createPlugin(const char* /*name*/, const PluginFieldCollection* fc) noexcept
    {
        if (!fc || !fc->fields) return nullptr;
        int32_t mnk[3] = {0, 0, 0};
        bool have[3] = {false, false, false};
        for (int32_t i = 0; i < fc->nbFields; ++i) {
            const PluginField& f = fc->fields[i];
            if (!f.name || !f.data || f.type != MIXQ_FIELD_INT32) continue;
            const std::string n(f.name);
            const int which = n == "m" ? 0 : n == "n" ? 1 : n == "k" ? 2 : -1;
            if (which >= 0) mnk[which] = *static_cast<const int32_t*>(f.data), have[which] = true;
        }
        if (!(have[0] && have[1] && have[2])) return nullptr;
        MixQPlugin* p = new (std::nothrow) MixQPlugin(mnk[0], mnk[1], mnk[2]);
        return finish(p);
    }
    MixQPlugin* deserializePlugin(const char* /*name*/, const void* serialData, std::size_t serialLength) noexcept /* .cpp:935-951 */
    {
        return finish(new (std::nothrow) MixQPlugin(serialData, serialLength));
    }
    void setPluginNamespace(const char* ns) noexcept { ns_ = ns ? ns : ""; }
    const char* getPluginNamespace() const noexcept { return ns_.c_str(); }

private:
    MixQPlugin* finish(MixQPlugin* p) noexcept
    {
        if (p && !p->valid()) {
            p->destroy();
            return nullptr;
        }
        if (p) p->setPluginNamespace(ns_.c_str());
        return p;
    }
    std::string ns_;
    PluginFieldCollection mFC{0, nullptr};
};

} // namespace mixq_plugin
#endif /* MIXQ_PLUGIN_HPP_ */
