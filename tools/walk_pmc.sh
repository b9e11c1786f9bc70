#!/usr/bin/env bash
# tools/walk_pmc.sh LIB.so [LIB.so ...] -- occupancy / wait / LDS counters of the backward blend of each build (two PMC passes each)
cd "$(dirname "$0")/.."
R="$PWD"; export TMPDIR=/tmp
for L in "$@"; do
  echo "== $L"
  export GSR_LIB_PATH="$(realpath "$L")"
  bash tools/pmc_kernel.sh blend_bwd "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY"
done
