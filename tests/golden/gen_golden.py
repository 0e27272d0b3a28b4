#!/usr/bin/env python3
"""Generate the committed golden fixtures from the importable pieces of the reference.

Runs ONLY in the build container (needs /root/reference); the GPU box and the test-suite read the
committed ``*.npz`` files and never this script's inputs.  Usage:  python tests/golden/gen_golden.py

What is captured (inputs + expected outputs, data only):
  pack_small.npz   W fp16 [96,512] (seeded), act-scale vector -> outputs of the REFERENCE functions
                   * ``to_quantized_weight`` (modelopt/torch/export/model_config_utils.py:298-308), executed
                     from the reference file itself (module loaded by path; its parent package __init__
                     needs the nvidia-modelopt dist-info, so the two modules are loaded standalone);
                   * the two expression lines of ``pack_linear_weights`` that need no CUDA/mixlib/EETQ
                     (:429-430 weights_scaling_factor, :446-448 fp_ind, :452-453 fp_weight + zeroing),
                     evaluated verbatim with torch on CPU.
  act_scales_llama.npz  three real activation-scale vectors from act_scales/Llama-2-1b.pt (layer-0 q_proj,
                   layer-0 up_proj, layer-1 down_proj [K=11008, max 1718]) and the reference selection
                   ``torch.sort(scales)[1][-128:]`` for each.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_model_config_utils():
    pkg = "modelopt.torch.export"
    for name in ("modelopt", "modelopt.torch", pkg):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []  # namespace-like parent so the relative import below resolves
            sys.modules[name] = m
    mods = {}
    for leaf in ("model_config", "model_config_utils"):
        path = f"{REF}/modelopt/torch/export/{leaf}.py"
        spec = importlib.util.spec_from_file_location(f"{pkg}.{leaf}", path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"{pkg}.{leaf}"] = mod
        spec.loader.exec_module(mod)
        mods[leaf] = mod
    return mods["model_config_utils"], mods["model_config"]


def main():
    mcu, mc = load_reference_model_config_utils()
    act = torch.load(f"{REF}/act_scales/Llama-2-1b.pt")

    # ---- act-scale vectors + reference outlier selection -------------------------------------
    keys = ["model.layers.0.self_attn.q_proj", "model.layers.0.mlp.up_proj", "model.layers.1.mlp.down_proj"]
    blob = {}
    for i, k in enumerate(keys):
        v = act[k].float().contiguous()
        blob[f"scales_{i}"] = v.numpy()
        blob[f"fp_ind_{i}"] = torch.sort(v)[1][-128:].to(torch.int32).numpy()  # model_config_utils.py:448
    blob["keys"] = np.array(keys)
    np.savez_compressed(os.path.join(OUT, "act_scales_llama.npz"), **blob)

    # ---- one small linear layer through the reference packing math ----------------------------
    g = torch.Generator().manual_seed(1234)
    N, K = 96, 512
    W = (torch.randn(N, K, generator=g) * 0.02).to(torch.float16)
    W[3, 7] = 0.5     # a dominant element: exercises the +-127 end of the grid
    W[5, :] = 0       # an all-zero row: scale 0 -> 0/0 -> NaN -> reference clamp/int8 behaviour
    layer_scales = act[keys[0]][:K].float().contiguous()
    weight = W.clone()
    # model_config_utils.py:429-430
    wsf = (torch.max(torch.abs(weight), dim=1)[0].unsqueeze(1) / (127)).to(torch.float16).reshape((weight.shape[0],))
    fp_ind = torch.sort(layer_scales)[1][-128:]                       # :446-448
    fp_weight = weight[:, fp_ind].clone()                             # :452
    weight[:, fp_ind] *= 0                                            # :453
    q = mcu.to_quantized_weight(weight, wsf, mc.QUANTIZATION_INT8_MIX)  # :460-464 -> reference function
    np.savez_compressed(
        os.path.join(OUT, "pack_small.npz"),
        W=W.numpy(), layer_scales=layer_scales.numpy(),
        weights_scaling_factor=wsf.numpy(), fp_ind=fp_ind.to(torch.int32).numpy(),
        fp_weight=fp_weight.numpy(), weight_int8=q.numpy(),
    )
    pflavour_fixture()
    model_walk_fixture(mcu, mc, act)
    print("wrote", os.listdir(OUT))


def model_walk_fixture(mcu, mc, act):
    """model_walk.npz: the REFERENCE's model-level walk -- ``merge_qkv`` (model_config_utils.py:203-217) then
    ``pack_linear_weights`` (:378-472), both executed from the reference file -- over two synthetic decoder layers
    (tests/golden/synth_model.py) held in the reference's own dataclasses (ModelConfig / DecoderLayerConfig / AttentionConfig /
    QKVConfig / MLPConfig / LinearConfig), driven by the real keys and vectors of act_scales/Llama-2-1b.pt.

    What the reference function needs that this container lacks, and how it is satisfied WHILE IT RUNS (nothing of it ships):
      * it opens the cwd-relative ``act_scales/Qwen2-72B.pt`` (:391): ``torch.load`` is pointed at a recording dict over
        Llama-2-1b.pt's tensors, which also captures the sequence of keys the walk reads (the key map);
      * ``mixlib.int8_matrix_to_half`` / ``mixlib.int_to_half`` (CUDA extension; bit-casts to fp16 carriers): bit-cast views;
      * ``EETQ.quant_weights`` (CUDA extension): a placeholder -- ``qweight`` / ``scales`` are NOT captured here;
      * ``Tensor.cuda`` -> identity.
    Captured per (layer, module): the act-scale key read, fp_ind (int32 view of the carrier), weights_scaling_factor, and
    SHA-256 digests of the int8 ``weight`` and of ``fp_weight`` (the tensors themselves are megabytes), plus the eight
    act-scale vectors the test needs as inputs."""
    import hashlib
    sys.path.insert(0, OUT)
    import synth_model as sm

    sd = sm.state_dict()

    def lin(name, layer):
        c = mc.LinearConfig()
        c.weight = sd[f"model.layers.{layer}.{name}.weight"].clone()
        return c

    layers = []
    for i in range(sm.LAYERS):
        qkv = mc.QKVConfig(q=lin("self_attn.q_proj", i), k=lin("self_attn.k_proj", i), v=lin("self_attn.v_proj", i))
        att = mc.AttentionConfig(qkv=qkv)
        mlp = mc.MLPConfig(fc=lin("mlp.gate_proj", i), gate=lin("mlp.up_proj", i), proj=lin("mlp.down_proj", i))  # layer_utils.py:753-786
        layers.append(mc.DecoderLayerConfig(attention=att, mlp=mlp))
    model_config = mc.ModelConfig(quantization=mc.QUANTIZATION_INT8_MIX, layers=layers)

    class Recording(dict):
        keys_read = []

        def __getitem__(self, k):
            Recording.keys_read.append(k)
            return dict.__getitem__(self, k)

    mixlib = types.ModuleType("mixlib")
    mixlib.int8_matrix_to_half = lambda t: t.contiguous().view(torch.float16)
    mixlib.int_to_half = lambda t: t.contiguous().view(torch.float16)
    eetq = types.ModuleType("EETQ")
    eetq.quant_weights = lambda w, dt, flag: (torch.zeros(w.shape, dtype=torch.int8), torch.zeros(w.shape[1], dtype=torch.float16))
    eetq.preprocess_weights = eetq.w8_a16_gemm = None
    keep = (torch.Tensor.cuda, torch.load, torch.cuda.empty_cache, sys.modules.get("mixlib"), sys.modules.get("EETQ"))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.load = lambda *a, **k: Recording(act)
    torch.cuda.empty_cache = lambda: None
    sys.modules["mixlib"], sys.modules["EETQ"] = mixlib, eetq
    try:
        mcu.merge_qkv(model_config)
        mcu.pack_linear_weights(model_config)
    finally:
        torch.Tensor.cuda, torch.load, torch.cuda.empty_cache = keep[0], keep[1], keep[2]
        for name, old in (("mixlib", keep[3]), ("EETQ", keep[4])):
            if old is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = old
    blob = {"keys_read": np.array(Recording.keys_read)}
    for i, dec in enumerate(model_config.layers):
        for which, cfg in (("attention.qkv", dec.attention.qkv), ("mlp.gate", dec.mlp.gate), ("mlp.proj", dec.mlp.proj)):
            tag = f"L{i}.{which}"
            w8 = cfg.weight.contiguous().view(torch.int8)          # the fp16 carrier [N, K/2] back to int8 [N, K]
            blob[f"{tag}.fp_ind"] = cfg.fp_ind.contiguous().view(torch.int32).numpy()
            blob[f"{tag}.weights_scaling_factor"] = cfg.weights_scaling_factor.numpy()
            blob[f"{tag}.weight_shape"] = np.array(w8.shape)
            blob[f"{tag}.weight_sha256"] = np.array(hashlib.sha256(w8.numpy().tobytes()).hexdigest())
            blob[f"{tag}.fp_weight_sha256"] = np.array(hashlib.sha256(cfg.fp_weight.contiguous().numpy().tobytes()).hexdigest())
            blob[f"{tag}.fp_weight_row0"] = cfg.fp_weight[0].contiguous().numpy()
            blob[f"{tag}.weight_row0"] = w8[0].numpy().copy()
    for i in range(sm.LAYERS):
        for name in ("self_attn.q_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"):
            blob[f"act.model.layers.{i}.{name}"] = act[f"model.layers.{i}.{name}"].float().contiguous().numpy()
    np.savez_compressed(os.path.join(OUT, "model_walk.npz"), **blob)


def pflavour_fixture():
    """pflavour_small.npz: outputs of the REFERENCE's own Python code for the PyTorch flavour
    (MixQ/src/mixquant/modules/linear.py), executed from the reference file on CPU tensors:
      * ``pack_to_i4`` (:13-17) on seeded int8 values;
      * ``MixLinear_GEMM.from_linear`` for bit = 8 (:113-120) and bit = 4 (:121-143): q_weight, scale_col, ind,
        weight_cache;
      * ``MixLinear_GEMM.FindOutliers`` (:155-161) on a seeded activation with inf / NaN / == sigma entries.
    The file imports the CUDA extension modules ``mixlib`` and ``EETQ`` at the top and calls ``.cuda()`` on tensors; for
    the import they are satisfied by empty placeholder modules and, while the functions run, ``Tensor.cuda`` /
    ``torch.cuda.get_device_capability`` are pointed at CPU no-ops -- none of the captured functions calls into either
    extension, every captured value is produced by the reference's own lines."""
    for name, attrs in (("mixlib", ()), ("EETQ", ("quant_weights", "preprocess_weights", "w8_a16_gemm"))):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, None)
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("ref_mixquant_linear", f"{REF}/MixQ/src/mixquant/modules/linear.py")
    lin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lin)

    keep_cuda, keep_cap = torch.Tensor.cuda, torch.cuda.get_device_capability
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.get_device_capability = lambda *a, **k: (8, 0)
    try:
        g = torch.Generator().manual_seed(4321)
        blob = {}
        q = torch.randint(-8, 8, (12, 64), generator=g, dtype=torch.int8)
        blob["i4_in"] = q.numpy()
        blob["i4_packed"] = lin.pack_to_i4(q).numpy()

        class Linear:  # the attributes from_linear reads from an nn.Linear
            pass

        class Cache:
            sigma = 6
            stop = 2

        N, K = 48, 512   # bit = 4 keeps the reference's default of 256 fp16 outlier columns
        W = (torch.randn(N, K, generator=g) * 0.02).to(torch.float16)
        W[2, 9] = 0.4
        layer_scales = torch.rand(K, generator=g)
        blob["W"], blob["layer_scales"] = W.numpy(), layer_scales.numpy()
        for bit in (8, 4):
            fake = Linear()
            fake.in_features, fake.out_features, fake.bias = K, N, None
            fake.weight = types.SimpleNamespace(data=W.clone())
            cache = Cache()
            cache.sigma = torch.tensor(6.0)
            ql = lin.MixLinear_GEMM.from_linear(fake, bit, cache=cache, layer_scales=layer_scales, dev="cpu")
            blob[f"w{bit}_q_weight"] = ql.q_weight.numpy()
            blob[f"w{bit}_scale_col"] = ql.scale_col.numpy().reshape(-1)
            if bit == 4:
                blob["w4_ind"] = ql.ind.numpy()
                blob["w4_weight_cache"] = ql.weight_cache.numpy()
        A = torch.randn(40, 256, generator=g).clamp_(-5, 5).to(torch.float16)
        A[torch.randint(0, 40, (25,), generator=g), torch.randint(0, 256, (25,), generator=g)] = 8.25
        A[3, 10], A[4, 11], A[5, 12] = float("nan"), float("inf"), 6.0
        finder = lin.MixLinear_GEMM.__new__(lin.MixLinear_GEMM)
        torch.nn.Module.__init__(finder)
        finder.sigma = torch.zeros((1, 1), dtype=torch.float16)
        finder.sigma[0] = 6
        blob["fo_A"] = A.numpy()
        blob["fo_ind"] = lin.MixLinear_GEMM.FindOutliers(finder, A).numpy()
    finally:
        torch.Tensor.cuda, torch.cuda.get_device_capability = keep_cuda, keep_cap
    np.savez_compressed(os.path.join(OUT, "pflavour_small.npz"), **blob)


def write_meta():
    import json
    json.dump({"torch": torch.__version__,
               "note": "the tie ORDER of fp_ind in act_scales_llama.npz / model_walk.npz is that of this build's (unstable) torch.sort on CPU -- "
                       "the reference's own call; tests compare the order only under the same torch version and column SETS otherwise"},
              open(os.path.join(OUT, "META.json"), "w"), indent=1)


if __name__ == "__main__":
    write_meta()
    main()
