#!/bin/bash
# fragment-stage variants: resident workgroups per CU, both kernel forms
for pp in "" 1; do
  for pc in 2 3 4 6; do
    echo "== per_pixel=${pp:-0} per_cu=$pc"
    if [ -n "$pp" ]; then export VF_RESOLVE_PER_PIXEL=1; else unset VF_RESOLVE_PER_PIXEL; fi
    VF_RESOLVE_PER_CU=$pc timeout -k 10 120 python tools/exp_fragment.py both || exit 1
  done
done
