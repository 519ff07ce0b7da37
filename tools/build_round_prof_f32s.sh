#!/usr/bin/env bash
# Diagnostic build of the float32-as-bf16-parts scan (hdb_mfma_f32s_b.hip) with per-wave section clocks (HDB_ROUND_PROF,
# hdb_mfma_kernel.h) -> tools/bin/libhyperdb_hip_rp.so (git-ignored); read by tools/round_sections.py.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
C="${HERE}/../local-hyperdb_amd/csrc"
mkdir -p "${HERE}/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -DHDB_ROUND_PROF=1 -c "${C}/hdb_mfma_f32s_b.hip" -o "${HERE}/bin/f32s_b_rp.o"
objs=$(ls "${C}"/obj/*.o | grep -v "hdb_mfma_f32s_b.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "${HERE}/bin/libhyperdb_hip_rp.so" ${objs} "${HERE}/bin/f32s_b_rp.o"
echo "built tools/bin/libhyperdb_hip_rp.so"
