#!/bin/bash
# HERE (build container): a variant build of the library for same-box A/B runs.  usage: build_variant.sh NAME "extra hipcc flags" [contract]
# -> ab/NAME.so (contract: on (default, the Makefile's) | fast)
NAME=$1; EXTRA=$2; CONTRACT=${3:-on}
R=$(cd $(dirname $0)/../.. && pwd)
D=/tmp/variant_$NAME; rm -rf $D; mkdir -p $D $R/ab
cd $R/ram-dsir_amd/csrc
ls *.hip | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=$CONTRACT -fno-slp-vectorize -fno-vectorize $EXTRA -w -c {} -o $D/\$(basename {} .hip).o" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/$NAME.so $D/*.o && echo "built ab/$NAME.so"
