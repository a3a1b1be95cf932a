/* The C ABI used from plain C (no Python, no C++): include/gtsfm_amd.h must compile as C99, the library must link, and the
 * host-only entry points must work without a GPU. Built and run by tests/test_abi_from_c.py with gcc. */
#include <stdio.h>
#include <string.h>

#include "gtsfm_amd.h"

int main(void) {
    int32_t first[8];
    int16_t weights[32];
    int sum, d, t;
    if (gtsfm_abi_version() <= 0) return 1;
    if (gtsfm_sp_packed_weight_floats() == 0) return 2;
    if (gtsfm_prep_cubic_taps(8, 16, first, weights) != GTSFM_OK) return 3; /* cv.resize(INTER_CUBIC) tap tables, host side */
    for (d = 0; d < 8; ++d) {
        sum = 0;
        for (t = 0; t < 4; ++t) sum += weights[4 * d + t];
        if (sum < 2046 || sum > 2050) return 4; /* 11-bit weights sum to ~2^11 */
        if (first[d] != 2 * d) return 5;        /* (d + 0.5) * 2 - 0.5 = 2 d + 0.5 */
    }
    if (gtsfm_prep_cubic_taps(0, 16, first, weights) != GTSFM_ERR_INVALID) return 6; /* argument errors are codes ... */
    if (strlen(gtsfm_last_error()) == 0) return 7;                                    /* ... plus a thread-local message */
    if (gtsfm_verify_workspace_bytes(1000) < 1000 * 36) return 8;
    /* device entry points reject null pointers before touching the GPU */
    if (gtsfm_verify_essential_f64(NULL, NULL, NULL, NULL, NULL, NULL, 10, NULL, NULL, 1.0, 1, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL) !=
        GTSFM_ERR_INVALID)
        return 9;
    printf("abi_from_c OK (ABI version %d)\n", gtsfm_abi_version());
    return 0;
}
