#!/usr/bin/env bash
# tools/ab3.sh A.so B.so [C.so] -- pairwise in-process A/B of builds against the first one
A="$1"; shift
for B in "$@"; do python tools/ab_lib.py "$A" "$B" 6 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['A'], d['median_A'], '|', d['B'], d['median_B'], '| delta us', round((d['median_B']-d['median_A'])*1e3,2))"; done
