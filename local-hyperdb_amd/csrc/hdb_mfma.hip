// hdb_mfma.hip -- batched Q.V^T scan on the matrix cores (placeholder until the MFMA kernel lands).
#include "hdb_common.h"
#include "../../include/hyperdb_hip.h"

extern "C" int hdb_mfma_supported(int dtype, int d, int metric) { (void)dtype; (void)d; (void)metric; return 0; }
extern "C" int hdb_launch_mfma_scan(const ScanArgs*, int, int, const void*, const float*, const float*, int, void*) {
    return (int)hipErrorNotSupported;
}
extern "C" int hdb_launch_q_to_f16(const float*, int, int, void*, void*) { return (int)hipErrorNotSupported; }
