#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG=..]...   ->  build/variants/libvf_NAME.so  (kernel experiments; compare with tools/quick_variants.sh)
name=$1; shift
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared "$@" vulkan_forge_amd/csrc/vf_hip.hip -o build/variants/libvf_$name.so
