#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG=..]...   ->  build/variants/libvf_NAME.so  (kernel experiments; compare with tools/quick_variants.sh)
name=$1; shift
mkdir -p build/variants
# (the flags of __graft_entry__.build(); VF_NO_TUNING=1 leaves the -mllvm code-generation switches out)
tuning=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIPCC_TUNING))")      # one list: the build's
[ -n "$VF_NO_TUNING" ] && tuning=""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $tuning "$@" vulkan_forge_amd/csrc/vf_hip.hip -o build/variants/libvf_$name.so && python tools/isa_lint.py build/variants/libvf_$name.so | grep -v ' 0 with the amount' ; true
