#!/bin/bash
# Builds pcs_icp.hip with -DPCS_WG_TRACE into the in-tree library (rebuild normally afterwards:
# python -m pcseqlearning_amd.build --force) and runs tools/wg_trace.py.  Run on the GPU box.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wno-unused-result"
/opt/rocm/bin/hipcc $F -DPCS_WG_TRACE -I include -c pcseqlearning_amd/csrc/pcs_icp.hip -o build/hip/pcs_icp_trace.o
objs=$(ls build/hip/*.o | grep -v "pcs_icp.o\|pcs_icp_trace.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pcseqlearning_amd/libpcs_hip.so $objs build/hip/pcs_icp_trace.o
python tools/wg_trace.py
