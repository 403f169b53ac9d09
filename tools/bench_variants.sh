#!/bin/bash
# One bench.py process per library variant (build/variants/libvf_*.so), one after another: the bench is repeatable to 0.2 %,
# which timing several handles inside one process (tools/exp_variants.py) is not.
for f in build/variants/libvf_*.so; do
  VF_HIP_LIB=$PWD/$f timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); o=d.get('other_camera') or {}; print('%-28s default %.4f ms   fill %.4f ms' % ('$(basename $f)', d['ms_per_step'], o.get('ms_per_step', 0)))"
done
