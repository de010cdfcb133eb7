/* Plain-C use of libqbp.so (include/qbp.h): decode the syndrome of main.py's example on the
 * Steane code.  gcc -std=c99 -Iinclude examples/decode_steane.c -Lqldpc_amd/csrc -lqbp -lm */
#include <math.h>
#include <stdio.h>
#include "qbp.h"

int main(void)
{
    /* Hx of generateCodeMatrices.py:64-68 in CSR form */
    const int32_t row_ptr[4] = {0, 4, 8, 12};
    const int32_t col_idx[12] = {0, 2, 4, 6, 1, 2, 5, 6, 3, 4, 5, 6};
    const uint8_t syndrome[3] = {1, 1, 0};             /* errors on qubits 0 and 1 (main.py:21-27) */
    double prior[7];
    uint8_t hard[7], converged;
    int32_t iteration;
    double llr[7];
    qbp_handle* h = NULL;
    for (int i = 0; i < 7; ++i) prior[i] = log((1 - 0.1) / 0.1);      /* main.py:17-18 */
    if (qbp_create(row_ptr, col_idx, 3, 7, 0, &h) != QBP_OK) {
        fprintf(stderr, "qbp_create: %s\n", qbp_last_error());
        return 1;
    }
    if (qbp_decode_batch(h, syndrome, prior, 1, 50, QBP_SUM_PRODUCT, 1.0, 1.0, 20.0, 0, hard, &converged,
                         &iteration, llr) != QBP_OK) {
        fprintf(stderr, "qbp_decode_batch: %s\n", qbp_last_error());
        return 1;
    }
    printf("converged %d at iteration %d: ", converged, iteration);
    for (int i = 0; i < 7; ++i) printf("%d", hard[i]);
    printf("\nllr:");
    for (int i = 0; i < 7; ++i) printf(" %.8f", llr[i]);
    printf("\n");
    qbp_destroy(h);
    return 0;
}
