#!/usr/bin/env bash
# tools/prof.sh TAG [bench args...] -- rocprofv3 kernel-trace + stats of bench.py on the GPU box.
# Writes gpurun_out/prof_TAG/ (CSV) and a compact per-kernel summary gpurun_out/prof_TAG_summary.txt
set -uo pipefail
TAG="$1"; shift
R="$PWD"
export TMPDIR=/tmp
mkdir -p "$R/gpurun_out"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$TAG" -o "$TAG" -- \
    python "$R/bench.py" "$@" > "$R/gpurun_out/prof_${TAG}_bench.log" 2>&1
cd "$R"
STATS=$(find "gpurun_out/prof_$TAG" -name "*kernel_stats.csv" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py $*"
  echo "# $(date -u) $(python -c 'import torch;print(torch.cuda.get_device_name(0))' 2>/dev/null)"
  grep '^{' "gpurun_out/prof_${TAG}_bench.log" | cut -c1-3000
  echo "# kernel stats (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev)"
  [ -n "$STATS" ] && head -25 "$STATS"
} > "gpurun_out/prof_${TAG}_summary.txt"
cat "gpurun_out/prof_${TAG}_summary.txt" | cut -c1-260
