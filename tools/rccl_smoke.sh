#!/usr/bin/env bash
# tools/rccl_smoke.sh -- first thing to run on a multi-GPU MI355X node: RCCL with N > 1 ranks has never executed on the
# builder's (single-GPU) boxes.  For N = 2, 4, 8 (up to the GPUs present): tools/rccl_check.py (ShardedAdam over RCCL against
# torch.optim.Adam, gradient sink, lazy all-gather) and a 10-step bench.py --gpus N line; then the GPU dist tests.
# Results land in gpurun_out/rccl_smoke/.
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/rccl_smoke; mkdir -p "$OUT"
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NGPU"
PORT=29611
for N in 2 4 8; do
  [ "$N" -le "$NGPU" ] || continue
  echo "== rccl_check N=$N"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT tools/rccl_check.py \
      > "$OUT/check_n$N.log" 2>&1; echo "rc=$? $(grep RCCL_CHECK_OK "$OUT/check_n$N.log" | tail -1)"
  PORT=$((PORT + 1))
  echo "== bench.py --gpus $N --steps 10"
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py \
      --gpus $N --steps 10 --warmup 3 > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"; echo "rc=$?"
  PORT=$((PORT + 1))
  python - "$OUT/bench_n$N.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  value", d["value"], "views/s  ms/step", d["ms_per_step"], " same step on one GPU:", d.get("same_step_on_one_gpu"))
except Exception as ex:
    print("  no bench line:", ex)
PY
done
echo "== bench.py --gpus 1 --scale-step (the N = 1 point of the same step)"
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --scale-step --no-cpu-baseline --no-extras > "$OUT/bench_n1_scale_step.json" 2> "$OUT/bench_n1.err"; echo "rc=$?"
echo "== tests/test_gpu_dist.py"
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -3
