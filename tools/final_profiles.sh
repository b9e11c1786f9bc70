#!/usr/bin/env bash
# tools/final_profiles.sh NAME -- everything behind profiles/NAME_*: kernel trace + stats, the PMC passes, pmc_latest.json, and
# (second step, because bench.py reads pmc_latest.json) the driver's own command line.  Run on the GPU box:
#   gpurun -- 'bash tools/final_profiles.sh r04_final'
NAME="${1:-final}"
bash tools/prof.sh fin --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-extras --views-in-flight 1 > /dev/null 2>&1
bash tools/pmc.sh fin > /dev/null 2>&1
python tools/summarize_profiles.py fin "$NAME" | tail -3
mkdir -p gpurun_out/profiles_out; cp profiles/${NAME}_kernel_stats.txt profiles/${NAME}_pmc.txt profiles/pmc_latest.json gpurun_out/profiles_out/
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/profiles_out/${NAME}_bench.json
cut -c1-400 gpurun_out/profiles_out/${NAME}_bench.json
