#!/usr/bin/env bash
# Builds libmixq_mi355x.so (HIP kernels + C ABI) for gfx950, in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libmixq_mi355x.so"
# one build at a time: two concurrent runs write the same objects (seen once: a library that aborted in its first new kernel)
exec 9>"${HERE}/.build.lock"
flock 9
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-fast-math -Wall -Wno-unused-function)
OBJS=()
PIDS=()
for f in quant_kernels gemm_kernels gemm_pp_kernels gemm_pp128_kernels gemm_mid_kernels gemm_skinny_kernels decode_kernels w8a16_gemm_kernels norm_kernels outlier_kernels int4_kernels int4_gemm_kernels tp_kernels mixq_api; do
  src="${HERE}/${f}.hip"; obj="${HERE}/${f}.o"
  if [[ ! -f "$obj" || "$src" -nt "$obj" || "${HERE}/mixq_device.h" -nt "$obj" || "${HERE}/mixq_launch.h" -nt "$obj" || "${HERE}/../../include/mixq.h" -nt "$obj" ]]; then
    rm -f "$obj"   # a failed compile must not leave a stale object behind
    "$HIPCC" "${FLAGS[@]}" ${EXTRA_HIPCC_FLAGS:-} -c "$src" -o "$obj" &
    PIDS+=($!)
  fi
  OBJS+=("$obj")
done
for pid in "${PIDS[@]:-}"; do
  [[ -z "$pid" ]] || wait "$pid" || { echo "build.sh: a compile job failed" >&2; exit 1; }
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}"
echo "built $OUT"
