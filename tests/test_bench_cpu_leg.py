"""CPU: the `cpu_baseline` leg of bench.py (the oracle timed on host cores) runs without a GPU and reports the contract
fields; the JSON-only-on-stdout plumbing is exercised on the GPU box by the bench itself."""
import importlib.util
import os

from conftest import ROOT


def test_cpu_baseline_leg_fields():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = bench.cpu_baseline(seconds_target=0.2)
    assert r["unit"] == "tokens/s" and r["kind"] == "port" and r["cores"] >= 1
    assert r["value"] > 0 and "oracle.linear_prefill" in r["sample"]
    # 8.99 GOP of int8 work per token from the model table the bench uses
    gop = sum(2.0 * n * k for _, n, k in bench.LLAMA2_7B["linears"]) * bench.LLAMA2_7B["layers"] / 1e9
    assert abs(gop - 8.9926) < 1e-3
    # the other two BASELINE models of the `configs` object (8.53 and 88.6 GOP / token; one GPU of TP = 8 carries an eighth)
    gq = sum(2.0 * n * k for _, n, k in bench.QWEN2_7B["linears"]) * bench.QWEN2_7B["layers"] / 1e9
    g70 = sum(2.0 * n * k for _, n, k in bench.LLAMA2_70B_TP8["linears"]) * bench.LLAMA2_70B_TP8["layers"] / 1e9
    assert abs(gq - 8.53) < 0.01 and abs(8 * g70 - 88.6) < 0.1
