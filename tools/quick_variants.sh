#!/bin/bash
# tools/exp_quick.py for the default library and every build/variants/libvf_*.so
timeout -k 10 200 python tools/exp_quick.py default || exit 1
for f in build/variants/libvf_*.so; do
  n=$(basename $f .so); VF_HIP_LIB=$PWD/$f timeout -k 10 200 python tools/exp_quick.py ${n#libvf_} || exit 1
done
