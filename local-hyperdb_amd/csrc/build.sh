#!/usr/bin/env bash
# Build libhyperdb_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../lib"
mkdir -p "${OUT}" "${HERE}/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed"
JOBS="${HDB_BUILD_JOBS:-$(nproc)}"
pids=()
for src in hdb_mfma_anyd_a hdb_mfma_anyd_b hdb_mfma_anyd_c hdb_mfma_anyd_d hdb_mfma_anyd_e hdb_mfma_anyd_f hdb_mfma_ksplit hdb_mfma_ksplit_s hdb_mfma_d384 hdb_mfma_f32 hdb_mfma_f32b hdb_mfma_f32s hdb_mfma_f32s_b hdb_mfma_qt2 hdb_mfma_wide hdb_mfma_mid hdb_mfma_narrow hdb_mfma_1k hdb_mfma_fused hdb_mfma_fused_wide hdb_bits_fused hdb_l1_tile hdb_scan hdb_select hdb_mfma hdb_sort hdb_rows hdb_api; do
  if [ ! -f "${HERE}/obj/${src}.o" ] || [ "${HERE}/${src}.hip" -nt "${HERE}/obj/${src}.o" ] || \
     [ "${HERE}/hdb_common.h" -nt "${HERE}/obj/${src}.o" ] || [ "${HERE}/hdb_mfma_kernel.h" -nt "${HERE}/obj/${src}.o" ] || [ "${HERE}/hdb_mfma_fused.h" -nt "${HERE}/obj/${src}.o" ] || [ "${HERE}/hdb_mfma_anyd.h" -nt "${HERE}/obj/${src}.o" ] || [ "${HERE}/hdb_finalize.h" -nt "${HERE}/obj/${src}.o" ] || [ "${HERE}/../../include/hyperdb_hip.h" -nt "${HERE}/obj/${src}.o" ]; then
    while [ "$(jobs -rp | wc -l)" -ge "${JOBS}" ]; do sleep 0.2; done
    ${HIPCC} ${FLAGS} -c "${HERE}/${src}.hip" -o "${HERE}/obj/${src}.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
${HIPCC} --offload-arch=gfx950 -shared -fPIC -o "${OUT}/libhyperdb_hip.so" \
  "${HERE}/obj/hdb_mfma_anyd_a.o" "${HERE}/obj/hdb_mfma_anyd_b.o" "${HERE}/obj/hdb_mfma_anyd_c.o" "${HERE}/obj/hdb_mfma_anyd_d.o" "${HERE}/obj/hdb_mfma_anyd_e.o" "${HERE}/obj/hdb_mfma_anyd_f.o" "${HERE}/obj/hdb_mfma_ksplit.o" "${HERE}/obj/hdb_mfma_ksplit_s.o" "${HERE}/obj/hdb_mfma_d384.o" "${HERE}/obj/hdb_mfma_f32.o" "${HERE}/obj/hdb_mfma_f32b.o" "${HERE}/obj/hdb_mfma_f32s.o" "${HERE}/obj/hdb_mfma_f32s_b.o" "${HERE}/obj/hdb_mfma_qt2.o" "${HERE}/obj/hdb_mfma_wide.o" "${HERE}/obj/hdb_mfma_mid.o" "${HERE}/obj/hdb_mfma_narrow.o" "${HERE}/obj/hdb_mfma_1k.o" "${HERE}/obj/hdb_mfma_fused.o" "${HERE}/obj/hdb_mfma_fused_wide.o" "${HERE}/obj/hdb_bits_fused.o" "${HERE}/obj/hdb_l1_tile.o" "${HERE}/obj/hdb_scan.o" "${HERE}/obj/hdb_select.o" "${HERE}/obj/hdb_mfma.o" "${HERE}/obj/hdb_sort.o" "${HERE}/obj/hdb_rows.o" "${HERE}/obj/hdb_api.o"
echo "built ${OUT}/libhyperdb_hip.so"
