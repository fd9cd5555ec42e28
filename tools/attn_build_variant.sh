#!/bin/bash
# attn_build_variant.sh <tag> <extra hipcc flags...>: builds ab_tmp/lib<tag>.so from the working tree with extra flags for attention.hip
tag=$1; shift
cd /root/repo/image_generator_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form "$@" -c attention.hip -o /tmp/attention_$tag.o || exit 1
objs=$(ls build/*.o | grep -v attention.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/attention_$tag.o -o /root/repo/ab_tmp/lib$tag.so
