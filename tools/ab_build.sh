#!/bin/bash
# A/B variants of one translation unit: recompiles <file.hip> with extra flags (diagnostic build) and links it with the
# other diagnostic objects into modelcrowdnav_amd/csrc/build_ab/<name>/libmcn_hip.so (load it with MCN_HIP_LIB).
#   bash tools/ab_build.sh <name> <file.hip> [-DFOO=1 ...]
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../modelcrowdnav_amd/csrc"
make -s diag > /dev/null
mkdir -p build_ab/$NAME
OBJ=${SRC%.hip}.o
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    -DMCN_DIAG "$@" -c $SRC -o build_ab/$NAME/$OBJ
OTHERS=$(ls build_diag/*.o | grep -v "/$OBJ$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_ab/$NAME/libmcn_hip.so build_ab/$NAME/$OBJ $OTHERS
echo "built build_ab/$NAME/libmcn_hip.so"
