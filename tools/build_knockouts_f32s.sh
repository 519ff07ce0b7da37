#!/usr/bin/env bash
# Measurement builds of the float32-as-bf16-parts scan (hdb_mfma_f32s_b.hip: d = 384 / 512 / 768) with parts of the round knocked out
# (HDB_MFMA_KNOCKOUT bits, hdb_mfma_kernel.h); results are wrong by design, only the time of a call means something.
# -> tools/bin/libhyperdb_hip_ko<bits>.so (git-ignored); run with HYPERDB_HIP_LIB=tools/bin/libhyperdb_hip_ko<bits>.so.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
C="${HERE}/../local-hyperdb_amd/csrc"
mkdir -p "${HERE}/bin"
for ko in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -DHDB_MFMA_KNOCKOUT=${ko} -c "${C}/hdb_mfma_f32s_b.hip" -o "${HERE}/bin/f32s_b_ko${ko}.o" &
done
wait
for ko in "$@"; do
  objs=$(ls "${C}"/obj/*.o | grep -v "hdb_mfma_f32s_b.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "${HERE}/bin/libhyperdb_hip_ko${ko}.so" ${objs} "${HERE}/bin/f32s_b_ko${ko}.o"
  echo "built tools/bin/libhyperdb_hip_ko${ko}.so"
done
