#!/bin/bash
# Build a diagnostic variant of libppf_hip.so with extra -D flags into build_var/NAME.so (select it with PPF_HIP_LIB).
#   tools/build_variant.sh NAME -DPPF_AGG_MIN_HITS=32 ...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build_var"
cd "$ROOT/yolo_ppf_pose_estimation_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" ppf_hip.hip -o "$ROOT/build_var/$NAME.so"
