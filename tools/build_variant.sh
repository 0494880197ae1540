#!/bin/bash
# usage: tools/build_variant.sh NAME FILE.hip "-DFLAGS"   -> scratch/NAME/libsonospec.so (other objects from csrc/build)
set -e
cd /root/repo/spectrogram_amd/csrc
name=$1; file=$2; flags=$3
mkdir -p ../../scratch/$name
obj=../../scratch/$name/$(basename $file .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -Wno-unused-result -ffp-contract=fast -fno-slp-vectorize $flags -c $file -o $obj
others=$(ls build/*.o | grep -v "build/$(basename $file .hip).o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scratch/$name/libsonospec.so $others $obj
echo built scratch/$name
