#!/bin/bash
# launch floors + in-kernel timelines of the small-M operator.  Output: gpurun_out/small_m_timeline.txt
cd "$(dirname "$0")/.."
O=gpurun_out/small_m_timeline.txt
{
  echo "## launch-floor probe (tools/experimental/launch_floor_probe.hip)"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/experimental/launch_floor_probe.hip -o /tmp/floor && /tmp/floor
  for m in 32 16 8; do
    echo; echo "## timeline M=$m 4096x4096"
    timeout 300 python tools/small_m_timeline.py --M $m
  done
  echo; echo "## timeline M=32 12288x4096"
  timeout 300 python tools/small_m_timeline.py --M 32 --N 12288
} > $O 2>&1
tail -100 $O
